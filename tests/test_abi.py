"""CPU: the C-ABI library builds for gfx950, loads, and exports every symbol include/dsm_hotpath.h
declares.  No compute calls (there is no GPU here and no CPU fallback)."""
import ctypes as C
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "dsm_hotpath.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(dsm_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_all_exported(built):
    from direct_stereo_slam_amd import _lib

    L = _lib.load()
    names = declared_symbols()
    assert len(names) >= 30
    for n in names:
        assert hasattr(L, n), f"{n} declared in include/dsm_hotpath.h but not exported"
    # and the Python binding table covers exactly the header
    assert sorted(_lib.SYMBOLS) == names


def test_library_contains_gfx950_code_object(built):
    from direct_stereo_slam_amd import _lib

    out = subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler", "--list", "--type=o", f"--input={_lib.LIB_PATH}"],
                         capture_output=True, text=True)
    blob = open(_lib.LIB_PATH, "rb").read()
    assert b"gfx950" in blob, "no gfx950 code object embedded"
    assert b"eval_kernel" in blob


def test_no_cpu_fallback(built):
    """without a GPU every compute entry point must fail loudly, not fall back"""
    import torch

    from direct_stereo_slam_amd import _lib

    L = _lib.load()
    assert L.dsm_abi_version() == _lib.ABI_VERSION == 4
    # the header's DSM_ABI_VERSION is what the library reports
    import re
    hdr = open(os.path.join(ROOT, 'include', 'dsm_hotpath.h')).read()
    assert int(re.search(r'#define DSM_ABI_VERSION (\d+)', hdr).group(1)) == L.dsm_abi_version()
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by the gpu tests")
    h = C.c_void_p()
    rc = L.dsm_context_create(0, C.byref(h))
    assert rc == -2 and not h.value  # DSM_ERR_NO_DEVICE
    assert b"no CPU fallback" in L.dsm_last_error()


def test_product_never_imports_the_oracle():
    """only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may touch oracle/"""
    pkg = os.path.join(ROOT, "direct_stereo_slam_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp", ".cpp", ".h")) or f == "Makefile":
                text = open(os.path.join(dirpath, f), errors="ignore").read()
                assert "dsm_oracle" not in text and "from oracle" not in text and "import oracle" not in text, \
                    f"{f} references the oracle"
    text = open(os.path.join(ROOT, "include", "dsm_hotpath.h")).read()
    assert "oracle" not in text.lower()


def test_host_side_sc_distance(built):
    """dsm_sc_distance / dsm_search_sc are host code by design (search_place.h:59-84): compare with
    the oracle and with a dense numpy dot product."""
    import numpy as np

    from direct_stereo_slam_amd import _lib
    from oracle import oracle as O

    L = _lib.load()
    rng = np.random.default_rng(0)
    sigs = []
    for _ in range(4):
        dense = np.where(rng.uniform(size=1200) < 0.4, rng.uniform(0.1, 2.0, 1200), 0.0)
        for s in range(60):  # per-sector L2 normalisation, ScanContext.cpp:137-141
            nrm = np.sqrt((dense[s * 20:(s + 1) * 20] ** 2).sum())
            if nrm > 0:
                dense[s * 20:(s + 1) * 20] /= nrm
        idx = np.nonzero(dense)[0].astype(np.int32)
        sigs.append((idx, dense[idx].copy(), dense))
    a = sigs[0]
    for b in sigs[1:]:
        d_lib = L.dsm_sc_distance(a[0].ctypes.data_as(_lib.c_int_p), a[1].ctypes.data_as(_lib.c_double_p), len(a[0]),
                                  b[0].ctypes.data_as(_lib.c_int_p), b[1].ctypes.data_as(_lib.c_double_p), len(b[0]), 60)
        d_orc = O.sc_distance(a[0], a[1], b[0], b[1], 60)
        assert d_lib == d_orc  # bit exact: same float accumulation
        assert abs(d_lib - (1 - a[2] @ b[2] / 60) / 2) < 1e-5
    # arg-min: strict '<' keeps the first minimal candidate
    cand = sigs[1:] + [sigs[1]]
    ids = (C.c_int * 4)(10, 11, 12, 13)
    ci = (_lib.c_int_p * 4)(*[c[0].ctypes.data_as(_lib.c_int_p) for c in cand])
    cv = (_lib.c_double_p * 4)(*[c[1].ctypes.data_as(_lib.c_double_p) for c in cand])
    cn = (C.c_int * 4)(*[len(c[0]) for c in cand])
    ri, rd = C.c_int(), C.c_float()
    assert L.dsm_search_sc(a[0].ctypes.data_as(_lib.c_int_p), a[1].ctypes.data_as(_lib.c_double_p), len(a[0]), 4, ids, ci, cv,
                           cn, 60, C.byref(ri), C.byref(rd)) == 0
    ds = [O.sc_distance(a[0], a[1], c[0], c[1], 60) for c in cand]
    assert ri.value == 10 + int(np.argmin(ds)) and rd.value == min(ds)


def _disassemble_gfx950(tmp_path):
    """{kernel symbol: [instruction lines]} of every gfx950 code object embedded in the library"""
    import shutil

    from direct_stereo_slam_amd import _lib

    so = os.path.join(tmp_path, "lib.so")
    shutil.copy(_lib.LIB_PATH, so)
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    subprocess.run([objdump, "--offloading", so], capture_output=True, text=True, check=True, cwd=tmp_path)
    funcs = {}
    for f in sorted(os.listdir(tmp_path)):
        if "gfx950" not in f:
            continue
        txt = subprocess.run([objdump, "-d", "--no-show-raw-insn", os.path.join(tmp_path, f)], capture_output=True, text=True, check=True).stdout
        cur = None
        for line in txt.splitlines():
            m = re.match(r"^[0-9a-f]+ <(.+)>:$", line)
            if m:
                cur = funcs.setdefault(m.group(1), [])
            elif cur is not None and line.strip() and not line.lstrip().startswith(("//", ";")):
                cur.append(line.split("//")[0].strip())
    return funcs


def test_ticket_atomics_follow_a_drained_store_queue(built, tmp_path):
    """The cross-workgroup hand-off (xwg_sync.hpp) is: device-scope stores, `s_waitcnt vmcnt(0)`, THEN the device-scope atomic
    that announces them.  Nothing in the language pins that order but the inline-assembly wait, so the emitted ISA is
    checked: in every kernel that hands data to another workgroup, no global atomic read-modify-write may be reached with
    a vector store still in flight (store ... atomic without a vmcnt(0) wait between them, in program order)."""
    funcs = _disassemble_gfx950(str(tmp_path))
    kernels = {n: ins for n, ins in funcs.items()
               if re.search(r"eval_kernelILi\dELb[01]ELb1E", n) or "queue_kernel" in n or "queue_seed_kernel" in n or "xwg_litmus_kernel" in n}
    assert any("queue_kernelILi0" in n for n in kernels) and any("xwg_litmus" in n for n in kernels)
    assert sum(1 for n in kernels if "eval_kernel" in n) >= 3  # the fused eval + LM kernels of the three modes
    checked = 0
    for name, ins in kernels.items():
        pending, n_atomics = None, 0
        for i, line in enumerate(ins):
            op = line.split()[0]
            # the payload of a hand-off is written with device-scope (sc1) stores; plain stores are private results
            if op.startswith(("global_store", "flat_store", "buffer_store")) and " sc1" in line:
                pending = (i, line)
            elif op == "s_waitcnt" and re.search(r"vmcnt\(0\)", line):
                pending = None
            elif op in ("s_branch", "s_endpgm", "s_setpc_b64"):  # what follows is not reached by falling through
                pending = None
            elif op.startswith(("global_atomic", "flat_atomic")):
                n_atomics += 1
                assert pending is None, f"{name}: `{line}` (instruction {i}) can overtake `{pending[1]}` (instruction {pending[0]}): no s_waitcnt vmcnt(0) between them"
        assert n_atomics >= 1, name
        checked += n_atomics
    assert checked >= 12
