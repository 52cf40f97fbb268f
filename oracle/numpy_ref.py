"""A SECOND, independent restatement of the tracker's hot path in vectorised numpy float32.  TEST INFRASTRUCTURE ONLY.

Written from the text of the reference (src/scale_optimization/TrackerAndScaler.cpp, ScaleAccumulator.h), NOT from
oracle/dsm_oracle.c: its job is to remove single-author risk from the C oracle (tests/test_oracle_numpy_ref.py asserts
the two agree -- per evaluation bit for bit, per LM run in the accept / reject sequence, the integer counts and the
result).  It does not make parity "pinned": the reference itself still cannot be built or run here.

Structure differs from the C oracle on purpose: whole-array numpy arithmetic (IEEE float32 element-wise = the reference's
scalar float code without FMA contraction), np.cumsum for the sequential float sums (quirk Q1: E in point order),
block-wise cumsum for the SSE lane accumulators with their 1k / 1m shift-up (quirk Q4), poses as 4x4 double matrices with
the Lie-group exponential taken by scipy.linalg.expm instead of Sophus' closed form, numpy.linalg for nothing that
decides (the LDLT is written out, pivoting on the largest |diagonal| as Eigen's LDLT does).

UPSTREAM-DSO semantics restated from their published form (not present under /root/reference, hence recalled, not cited):
getInterpolatedElement33 (bilinear, weights dxdy, dy-dxdy, dx-dxdy, 1-dx-dy+dxdy on taps (+1,+1),(0,+1),(+1,0),(0,0)),
AffLight::fromToVecExposure, Accumulator9's lane / shift-up scheme (the same scheme ScaleAccumulator.h:60-105 spells out in
the reference tree), Eigen's Matrix3f::inverse() (cofactors * 1/det) and quaternion -> rotation matrix.
"""
import math

import numpy as np

f32 = np.float32


def _f(x):
    return np.asarray(x, dtype=np.float32)


def quat_to_rot(q):
    """Eigen QuaternionBase::toRotationMatrix (double)"""
    x, y, z, w = (float(v) for v in q)
    tx, ty, tz = 2.0 * x, 2.0 * y, 2.0 * z
    twx, twy, twz = tx * w, ty * w, tz * w
    txx, txy, txz = tx * x, ty * x, tz * x
    tyy, tyz, tzz = ty * y, tz * y, tz * z
    return np.array([[1.0 - (tyy + tzz), txy - twz, txz + twy],
                     [txy + twz, 1.0 - (txx + tzz), tyz - twx],
                     [txz - twy, tyz + twx, 1.0 - (txx + tyy)]])


def pose_to_matrix(pose):
    T = np.eye(4)
    T[:3, :3] = quat_to_rot(pose[:4])
    T[:3, 3] = pose[4:7]
    return T


def matrix_to_pose(T):
    """rotation matrix -> unit quaternion (x, y, z, w), w >= 0 ; translation"""
    from scipy.spatial.transform import Rotation

    q = Rotation.from_matrix(T[:3, :3]).as_quat()
    if q[3] < 0:
        q = -q
    return np.concatenate([q, T[:3, 3]])


def se3_exp(xi):
    """SE3::exp of the tangent (upsilon, omega) as a 4x4 matrix: the matrix exponential of the twist"""
    from scipy.linalg import expm

    ups, om = xi[:3], xi[3:6]
    A = np.zeros((4, 4))
    A[:3, :3] = [[0, -om[2], om[1]], [om[2], 0, -om[0]], [-om[1], om[0], 0]]
    A[:3, 3] = ups
    return expm(A)


def aff_from_to(expF, expT, g2F, g2T):
    """AffLight::fromToVecExposure (upstream DSO; call sites TrackerAndScaler.cpp:647-649,717-720)"""
    expF, expT = float(f32(expF)), float(f32(expT))
    if expF == 0 or expT == 0:
        expF = expT = 1.0
    a = math.exp(g2T[0] - g2F[0]) * expT / expF
    b = g2T[1] - a * g2F[1]
    return a, b


def mat3f_inverse(K):
    """Eigen Matrix3f::inverse(): every entry = cofactor * (1 / det), float32 (call site TrackerAndScaler.cpp:139)"""
    K = _f(K)

    def cof(i, j):
        a, b = [r for r in range(3) if r != i], [c for c in range(3) if c != j]
        m = f32(K[a[0], b[0]] * K[a[1], b[1]]) - f32(K[a[0], b[1]] * K[a[1], b[0]])
        return f32(m) if (i + j) % 2 == 0 else f32(-m)

    c0 = [cof(0, 0), cof(1, 0), cof(2, 0)]  # cofactors of column 0
    det = f32(f32(f32(c0[0] * K[0, 0]) + f32(c0[1] * K[1, 0])) + f32(c0[2] * K[2, 0]))
    invdet = f32(f32(1.0) / det)
    out = np.zeros((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            out[i, j] = f32(cof(j, i) * invdet)  # inverse = adjugate / det, adjugate(i,j) = cofactor(j,i)
    return out


def mat3f_mul(A, B):
    """3x3 float product as a coefficient-wise lazy product evaluates it: ((a0 b0 + a1 b1) + a2 b2)"""
    A, B = _f(A), _f(B)
    out = np.zeros((3, 3), np.float32)
    for i in range(3):
        for j in range(3):
            out[i, j] = f32(f32(f32(A[i, 0] * B[0, j]) + f32(A[i, 1] * B[1, j])) + f32(A[i, 2] * B[2, j]))
    return out


def interp33(img, x, y):
    """getInterpolatedElement33 on an (h, w, 3) float32 image for float32 coordinate arrays"""
    ix, iy = x.astype(np.int32), y.astype(np.int32)
    dx, dy = x - ix.astype(np.float32), y - iy.astype(np.float32)
    dxdy = dx * dy
    w11, w01, w10, w00 = dxdy, dy - dxdy, dx - dxdy, f32(1) - dx - dy + dxdy
    t00, t10, t01, t11 = img[iy, ix], img[iy, ix + 1], img[iy + 1, ix], img[iy + 1, ix + 1]
    return ((w11[:, None] * t11 + w01[:, None] * t01) + w10[:, None] * t10) + w00[:, None] * t00


def seq_sum(terms):
    """sequential float32 sum in array order (what `float E = 0; for (...) E += term;` computes)"""
    terms = _f(terms)
    return f32(0) if len(terms) == 0 else np.cumsum(terms, dtype=np.float32)[-1]


def lane_accumulate(prod):
    """prod: (n,) float32 per-entry products, n % 4 == 0, in buffer order.  The reference's accumulators keep 4 SSE lanes
    (entry i goes to lane i % 4), add pack after pack, and after every 1001st pack move the lane sums to the "1k" buffer --
    which, holding num_in_1k = 1001 > 1000 packs, is forwarded to the "1m" buffer at once (ScaleAccumulator.h:85-105, as
    written); finish() forces a last shift and adds the four lanes left to right (:43-57)."""
    lanes = _f(prod).reshape(-1, 4)
    d1m = np.zeros(4, np.float32)
    for b0 in range(0, len(lanes), 1001):
        blk = lanes[b0:b0 + 1001]
        d = np.cumsum(blk, axis=0, dtype=np.float32)[-1]  # SSEData: += pack after pack
        d1k = d + np.zeros(4, np.float32)                  # shiftUp: 1k = data + 1k (1k was zero)
        d1m = d1k + d1m                                    # ... and straight on: 1m = 1k + 1m
    return f32(f32(f32(d1m[0] + d1m[1]) + d1m[2]) + d1m[3])


def ldlt_solve(A, b):
    """Eigen LDLT<MatrixXd, Lower>::solve: LDL^T with symmetric pivoting on the largest |diagonal| (first one on ties),
    D^-1 with Eigen's tolerance, in double"""
    A = np.array(A, np.float64)
    n = len(A)
    A = np.tril(A) + np.tril(A, -1).T  # only the lower triangle is read
    perm = list(range(n))
    for k in range(n):
        p = k + int(np.argmax(np.abs(np.diag(A)[k:])))
        if p != k:
            A[[k, p], :] = A[[p, k], :]
            A[:, [k, p]] = A[:, [p, k]]
            perm[k], perm[p] = perm[p], perm[k]
        d = A[k, k]
        if k == 0 and abs(d) == 0.0 and not np.any(A):
            return np.zeros(n)
        if abs(d) > 0.0:
            l = A[k + 1:, k] / d
            A[k + 1:, k + 1:] -= np.outer(l, A[k + 1:, k])
            A[k + 1:, k] = l
    L = np.tril(A, -1) + np.eye(n)
    D = np.diag(A).copy()
    y = np.array(b, np.float64)[perm]
    for i in range(n):  # L y' = y
        y[i] -= L[i, :i] @ y[:i]
    tol = 1.0 / np.finfo(np.float64).max
    y = np.where(np.abs(D) > tol, y / np.where(D == 0, 1.0, D), 0.0)
    for i in range(n - 1, -1, -1):  # L^T x' = y'
        y[i] -= L[i + 1:, i] @ y[i + 1:]
    x = np.zeros(n)
    x[perm] = y
    return x


class NumpyTracker:
    """TrackerAndScaler (TrackerAndScaler.h:34-137) in numpy; parameter names follow dsm_params / orc_params"""

    def __init__(self, w, h, nlevels, T_f1_f0, K1, huber=9.0, cutoff=20.0, scale_xi_rot=1.0, scale_xi_trans=0.5, scale_a=10.0,
                 scale_b=1000.0, mode_a=0.0, mode_b=0.0, lambda_limit=0.001, max_iterations=(10, 20, 50, 50, 50, 50)):
        self.nl = nlevels
        self.w = [w >> l for l in range(nlevels)]
        self.h = [h >> l for l in range(nlevels)]
        self.huber, self.cutoff0 = f32(huber), f32(cutoff)
        self.scales = np.array([scale_xi_rot] * 3 + [scale_xi_trans] * 3 + [scale_a, scale_b], np.float64)  # :541-545 / :685-696
        self.mode_a, self.mode_b, self.lambda_limit = f32(mode_a), f32(mode_b), f32(lambda_limit)
        self.max_iterations = list(max_iterations)
        self.T10 = np.array(T_f1_f0, np.float64).reshape(4, 4)  # tfm_f1_f0_ (:82-86)
        # camera 1 pyramid (:89-98): float storage, double arithmetic in the expressions with 0.5
        self.fx1, self.fy1, self.cx1, self.cy1 = ([f32(v)] for v in K1)
        for l in range(1, nlevels):
            self.fx1.append(f32(float(self.fx1[l - 1]) * 0.5))
            self.fy1.append(f32(float(self.fy1[l - 1]) * 0.5))
            self.cx1.append(f32((float(self.cx1[0]) + 0.5) / (1 << l) - 0.5))
            self.cy1.append(f32((float(self.cy1[0]) + 0.5) / (1 << l) - 0.5))
        self.res_evals = [0] * nlevels
        self.trace = []

    def make_k(self, fx, fy, cx, cy):  # :117-141
        self.fx, self.fy, self.cx, self.cy = [f32(fx)], [f32(fy)], [f32(cx)], [f32(cy)]
        for l in range(1, self.nl):
            self.fx.append(f32(float(self.fx[l - 1]) * 0.5))
            self.fy.append(f32(float(self.fy[l - 1]) * 0.5))
            self.cx.append(f32((float(self.cx[0]) + 0.5) / (1 << l) - 0.5))
            self.cy.append(f32((float(self.cy[0]) + 0.5) / (1 << l) - 0.5))
        self.Ki = [mat3f_inverse([[self.fx[l], 0, self.cx[l]], [0, self.fy[l], self.cy[l]], [0, 0, 1]]) for l in range(self.nl)]

    def set_ref(self, ref_a, ref_b, ref_exposure, pc_u, pc_v, pc_idepth, pc_color):  # the result of setCoarseTrackingRef :317-327
        self.ref_aff, self.ref_exposure = (float(ref_a), float(ref_b)), ref_exposure
        self.pc = [tuple(_f(a[l]) for a in (pc_u, pc_v, pc_idepth, pc_color)) for l in range(self.nl)]

    def scale_depth(self, s):  # :329-336
        self.pc = [(u, v, (idp / f32(s)).astype(np.float32), c) for u, v, idp, c in self.pc]

    def set_frame(self, slot, dIp, exposure):
        if slot == 0:
            self.new_dIp, self.new_exposure = [_f(a) for a in dIp], exposure
        else:
            self.right_dIp, self.right_exposure = [_f(a) for a in dIp], exposure

    # ---- calcResPose :699-852 / calcResScale :1007-1172: shared warp-and-gather core ----------------------------------
    def _warp(self, lvl, M, t, img, fx, fy, cx, cy, cutoff, aff, flow_Ki=None, flow_M=None):
        u_, v_, id_, col = self.pc[lvl]
        x, y = u_, v_
        wl, hl = self.w[lvl], self.h[lvl]
        pt = [((M[r, 0] * x + M[r, 1] * y) + M[r, 2]) + t[r] * id_ for r in range(3)]  # :747 / :1061
        with np.errstate(divide="ignore", invalid="ignore"):
            u, v = pt[0] / pt[2], pt[1] / pt[2]
            new_id = id_ / pt[2]
        Ku, Kv = fx * u + cx, fy * v + cy
        flow = np.zeros(3, np.float32)
        if lvl == 0 and flow_Ki is not None:  # :754-784 / :1070-1100: every 32nd template index
            s = slice(0, None, 32)
            xs, ys, ids = x[s], y[s], id_[s]
            kx = [(flow_Ki[r, 0] * xs + flow_Ki[r, 1] * ys) + flow_Ki[r, 2] for r in range(3)]
            rx = [(flow_M[r, 0] * xs + flow_M[r, 1] * ys) + flow_M[r, 2] for r in range(3)]
            tid = [t[r] * ids for r in range(3)]

            def proj(p):
                return fx * (p[0] / p[2]) + cx, fy * (p[1] / p[2]) + cy

            with np.errstate(divide="ignore", invalid="ignore"):
                KuT, KvT = proj([kx[r] + tid[r] for r in range(3)])
                KuT2, KvT2 = proj([kx[r] - tid[r] for r in range(3)])
                Ku3, Kv3 = proj([rx[r] - tid[r] for r in range(3)])
            sq = lambda a, b: (a - xs) * (a - xs) + (b - ys) * (b - ys)
            flow[0] = seq_sum(np.stack([sq(KuT, KvT), sq(KuT2, KvT2)], 1).ravel())
            flow[1] = seq_sum(np.stack([sq(Ku[s], Kv[s]), sq(Ku3, Kv3)], 1).ravel())
            flow[2] = seq_sum(np.full(len(xs), 2, np.float32))
        with np.errstate(invalid="ignore"):
            inb = (Ku > 2) & (Kv > 2) & (Ku < f32(wl - 3)) & (Kv < f32(hl - 3)) & (new_id > 0)  # :786 / :1102
        idx = np.nonzero(inb)[0]
        hit = interp33(img, Ku[idx], Kv[idx])  # :790 / :1106
        fin = np.isfinite(hit[:, 0])
        idx, hit = idx[fin], hit[fin]
        refc = col[idx]
        if aff is not None:
            residual = hit[:, 0] - (f32(aff[0]) * refc + f32(aff[1]))  # :793
        else:
            residual = hit[:, 0] - refc  # :1109
        ar = np.abs(residual)
        with np.errstate(divide="ignore"):
            hw = np.where(ar < self.huber, f32(1), self.huber / ar).astype(np.float32)  # :794-795
        sat = ar > cutoff  # :797
        max_energy = f32(f32(f32(2) * self.huber) * cutoff) - f32(self.huber * self.huber)  # :726-728
        terms = np.where(sat, max_energy, ((hw * residual) * residual) * (f32(2) - hw)).astype(np.float32)  # :800 / :809
        E = seq_sum(terms)
        n_terms, n_sat = len(idx), int(sat.sum())
        keep = ~sat
        self.res_evals[lvl] += 1
        with np.errstate(invalid="ignore", divide="ignore"):
            rs = np.array([E, n_terms, float(flow[0]) / (float(flow[2]) + 0.1), 0.0, float(flow[1]) / (float(flow[2]) + 0.1),
                           f32(n_sat) / f32(n_terms) if n_terms else np.nan])  # :843-851
        buf = dict(idx=idx[keep], u=u[idx][keep], v=v[idx][keep], new_id=new_id[idx][keep], dx=hit[keep, 1], dy=hit[keep, 2],
                   residual=residual[keep], hw=hw[keep], refc=refc[keep])
        return rs, buf

    @staticmethod
    def _pad4(a):  # :824-834: zero entries up to a multiple of 4 (they count in n: quirk Q3)
        a = _f(a)
        return np.concatenate([a, np.zeros((-len(a)) % 4, np.float32)])

    def calc_res_pose(self, lvl, T, aff, cutoff):
        R = T[:3, :3].astype(np.float32)
        RKi = mat3f_mul(R, self.Ki[lvl])  # :715
        t = T[:3, 3].astype(np.float32)   # :716
        a, b = aff_from_to(self.ref_exposure, self.new_exposure, self.ref_aff, aff)  # :717-720
        rs, buf = self._warp(lvl, RKi, t, self.new_dIp[lvl], self.fx[lvl], self.fy[lvl], self.cx[lvl], self.cy[lvl], f32(cutoff),
                             (a, b), self.Ki[lvl], RKi)
        self.pose_buf = buf
        return rs

    def calc_gs_pose(self, lvl, aff):  # calcGSSSEPose :640-697
        B = self.pose_buf
        fxl, fyl = self.fx[lvl], self.fy[lvl]
        a = f32(aff_from_to(self.ref_exposure, self.new_exposure, self.ref_aff, aff)[0])
        b0 = f32(self.ref_aff[1])
        pad = self._pad4
        dx, dy = pad(B["dx"]) * fxl, pad(B["dy"]) * fyl  # :658-659
        u, v, idp = pad(B["u"]), pad(B["v"]), pad(B["new_id"])
        zero, one = f32(0), f32(1)
        J = [idp * dx, idp * dy, zero - idp * (u * dx + v * dy), zero - ((u * v) * dx + dy * (one + v * v)),
             (u * v) * dy + dx * (one + u * u), u * dy - v * dx, a * (b0 - pad(B["refc"])), np.full(len(u), -1, np.float32),
             pad(B["residual"])]  # :664-678
        wgt = pad(B["hw"])
        n = len(u)
        Hf = np.zeros((9, 9), np.float32)
        for r in range(9):
            Jw = J[r] * wgt
            for c in range(r, 9):
                Hf[r, c] = Hf[c, r] = lane_accumulate(Jw * J[c]) if n else f32(0)
        with np.errstate(divide="ignore", invalid="ignore"):
            invn = f32(1.0) / f32(n)
            H = Hf[:8, :8].astype(np.float64) * float(invn)  # :682-683
            b = Hf[:8, 8].astype(np.float64) * float(invn)
        H = (H * self.scales[None, :]) * self.scales[:, None]  # :685-692
        return H, b * self.scales, n

    def calc_res_scale(self, lvl, scale, cutoff):
        R10 = self.T10[:3, :3].astype(np.float32)
        M = mat3f_mul(R10, self.Ki[lvl])  # :1022-1023
        t = self.T10[:3, 3].astype(np.float32)
        S = (f32(scale) * M).astype(np.float32)  # `scale * rot_f1_f0_K0_i` (:1061)
        rs, buf = self._warp(lvl, S, t, self.right_dIp[lvl], self.fx1[lvl], self.fy1[lvl], self.cx1[lvl], self.cy1[lvl], f32(cutoff),
                             None, (f32(scale) * self.Ki[lvl]).astype(np.float32), S)
        u_, v_, id_, _ = self.pc[lvl]
        i = buf["idx"]
        with np.errstate(divide="ignore", invalid="ignore"):
            buf["rx"] = [(((M[r, 0] * u_[i] + M[r, 1] * v_[i]) + M[r, 2]) / id_[i]).astype(np.float32) for r in range(3)]  # :1068
        self.scale_buf = buf
        return rs

    def calc_gs_scale(self, lvl, scale):  # calcGSSSEScale :966-1005
        B = self.scale_buf
        pad = self._pad4
        t = self.T10[:3, 3].astype(np.float32)
        tx, ty, tz, s, one = t[0], t[1], t[2], f32(scale), f32(1)
        dxfx, dyfy = pad(B["dx"]) * self.fx1[lvl], pad(B["dy"]) * self.fy1[lvl]
        rx1, rx2, rx3 = (pad(a) for a in B["rx"])
        deno_sqrt = s * rx3 + tz
        with np.errstate(divide="ignore", invalid="ignore"):
            deno = one / (deno_sqrt * deno_sqrt)
            xno, yno = rx1 * tz - rx3 * tx, rx2 * tz - rx3 * ty
            J0 = dxfx * (deno * xno) + dyfy * (deno * yno)
        J1, w = pad(B["residual"]), pad(B["hw"])
        n = len(J0)
        J0w = J0 * w
        h00, h01 = (lane_accumulate(J0w * J0), lane_accumulate(J0w * J1)) if n else (f32(0), f32(0))
        with np.errstate(divide="ignore", invalid="ignore"):
            invn = f32(1.0) / f32(n)
            return f32(h00 * invn), f32(h01 * invn), n  # :1003-1004

    # ---- trackNewestCoarse :451-638 -----------------------------------------------------------------------------------
    def track(self, pose, aff, coarsest, min_res=None):
        T = pose_to_matrix(np.asarray(pose, np.float64))
        aff = (float(aff[0]), float(aff[1]))
        min_res = [np.nan] * 6 if min_res is None else list(min_res)
        last = [np.nan] * 6  # :459
        flow = [1000.0] * 3  # :460
        have_repeated = False
        self.res_evals = [0] * self.nl
        self.trace = []
        lvl = coarsest
        while lvl >= 0:
            rep = f32(1)
            res_old = self.calc_res_pose(lvl, T, aff, self.cutoff0 * rep)  # :475
            while res_old[5] > 0.6 and rep < 50:  # :477-485
                rep = f32(rep * 2)
                res_old = self.calc_res_pose(lvl, T, aff, self.cutoff0 * rep)
            H, b, _ = self.calc_gs_pose(lvl, aff)  # :487
            lam = f32(0.01)
            for _ in range(self.max_iterations[lvl]):  # :505
                Hl = H.copy()
                Hl[np.diag_indices(8)] *= (1 + float(lam))  # :506-508
                inc = ldlt_solve(Hl, -b)  # :509
                if self.mode_a < 0 and self.mode_b < 0:  # :511-515
                    inc = np.concatenate([ldlt_solve(Hl[:6, :6], -b[:6]), [0, 0]])
                if not (self.mode_a < 0) and self.mode_b < 0:  # :516-520
                    inc = np.concatenate([ldlt_solve(Hl[:7, :7], -b[:7]), [0]])
                if self.mode_a < 0 and not (self.mode_b < 0):  # :521-534
                    Hs, bs = Hl.copy(), b.copy()
                    Hs[:, 6] = Hs[:, 7]
                    Hs[6, :] = Hs[7, :]
                    bs[6] = bs[7]
                    st = ldlt_solve(Hs[:7, :7], -bs[:7])
                    inc = np.concatenate([st[:6], [0, st[6]]])
                extrap = f32(1)
                if lam < self.lambda_limit:  # :536-539
                    extrap = f32(math.sqrt(math.sqrt(float(f32(self.lambda_limit / lam)))))
                inc = inc * float(extrap)
                inc_scaled = inc * self.scales  # :541-545
                if not np.isfinite(inc_scaled.sum()):
                    inc_scaled[:] = 0  # :547-548
                T_new = se3_exp(inc_scaled[:6]) @ T  # :550-551
                aff_new = (aff[0] + inc_scaled[6], aff[1] + inc_scaled[7])  # :552-554
                res_new = self.calc_res_pose(lvl, T_new, aff_new, self.cutoff0 * rep)  # :556
                accept = (res_new[0] / res_new[1]) < (res_old[0] / res_old[1])  # :559
                self.trace.append((lvl, bool(accept)))
                if accept:  # :576-581
                    H, b, _ = self.calc_gs_pose(lvl, aff_new)
                    res_old, aff, T = res_new, aff_new, T_new
                    lam = f32(lam * f32(0.5))
                else:  # :583-585
                    lam = f32(lam * f32(4))
                    if lam < self.lambda_limit:
                        lam = self.lambda_limit
                if not (np.linalg.norm(inc) > 1e-3):  # :588
                    break
            last[lvl] = float(np.sqrt(f32(res_old[0] / res_old[1])))  # :596
            flow = list(res_old[2:5])  # :597
            if last[lvl] > 1.5 * min_res[lvl]:  # :598
                return False, matrix_to_pose(T), aff, last, flow
            if rep > 1 and not have_repeated:  # :601-604
                lvl += 1
                have_repeated = True
            lvl -= 1
        good = True
        if (self.mode_a != 0 and abs(f32(aff[0])) > 1.2) or (self.mode_b != 0 and abs(f32(aff[1])) > 200):  # :615-617
            good = False
        rel = aff_from_to(self.ref_exposure, self.new_exposure, self.ref_aff, aff)  # :619-622
        if (self.mode_a == 0 and abs(math.log(float(f32(rel[0])))) > 1.5) or (self.mode_b == 0 and abs(f32(rel[1])) > 200):  # :624-626
            good = False
        return good, matrix_to_pose(T), aff, last, flow

    # ---- optimizeScale :854-964 ---------------------------------------------------------------------------------------
    def optimize_scale(self, scale, coarsest):
        cur = f32(scale)
        last = [np.nan] * 6
        have_repeated = False
        self.res_evals = [0] * self.nl
        self.trace = []
        lvl = coarsest
        while lvl >= 0:
            rep = f32(1)
            res_old = self.calc_res_scale(lvl, cur, self.cutoff0 * rep)  # :873
            while res_old[5] > 0.6 and rep < 50:  # :875-883
                rep = f32(rep * 2)
                res_old = self.calc_res_scale(lvl, cur, self.cutoff0 * rep)
            H, b, _ = self.calc_gs_scale(lvl, cur)  # :885
            lam = f32(0.01)
            for _ in range(self.max_iterations[lvl]):
                Hl = f32(H * f32(1 + lam))  # :897-898
                with np.errstate(divide="ignore", invalid="ignore"):
                    inc = f32(-b / Hl)  # :899
                extrap = f32(1)
                if lam < self.lambda_limit:  # :901-904
                    extrap = f32(math.sqrt(math.sqrt(float(f32(self.lambda_limit / lam)))))
                inc = f32(inc * extrap)
                if not np.isfinite(inc) or abs(inc) > cur:  # :906-907
                    inc = f32(0)
                new = f32(cur + inc)  # :909
                res_new = self.calc_res_scale(lvl, new, self.cutoff0 * rep)  # :911
                accept = (res_new[0] / res_new[1]) < (res_old[0] / res_old[1])  # :914
                self.trace.append((lvl, bool(accept)))
                if accept:  # :926-930
                    H, b, _ = self.calc_gs_scale(lvl, new)
                    res_old, cur = res_new, new
                    lam = f32(lam * f32(0.5))
                else:  # :931-935
                    lam = f32(lam * f32(4))
                    if lam < self.lambda_limit:
                        lam = self.lambda_limit
                if not (inc > 1e-3):  # :937 -- SIGNED (quirk Q7)
                    break
            last[lvl] = float(np.sqrt(f32(res_old[0] / res_old[1])))  # :945
            if rep > 1 and not have_repeated:  # :947-950
                lvl += 1
                have_repeated = True
            lvl -= 1
        return last[0], float(cur)  # :954, :963
