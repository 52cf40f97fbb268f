#!/bin/bash
# How does FETCH_SIZE tally the two read streams of the eval kernels?  Known byte counts (tools/microbench/gather_pattern):
#   mode 0  = the template stream alone (16 B per point, one global_load_dwordx4 per lane),
#   mode 30 = template + the twelve-intensity taps of the 4-byte-per-texel target, no arithmetic.
# bash tools/pmc_calibrate.sh rNN   (through gpurun) -> gpurun_out/<tag>_profiles/<tag>_pmc_calibration.json
set -u
TAG=${1:-r02}
R=${GRAFT_REPO_ROOT:-$PWD}
OUT=$R/gpurun_out/${TAG}_cal
mkdir -p $OUT $R/gpurun_out/${TAG}_profiles
hipcc --offload-arch=gfx950 -O3 $R/tools/microbench/gather_pattern.hip -o /tmp/gather 2>/dev/null
cd /tmp && export TMPDIR=/tmp
for m in 0 30; do
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $OUT/m$m -- /tmp/gather 256 $m $m > $OUT/m$m.log 2>&1
done
python - "$OUT" "$R/gpurun_out/${TAG}_profiles/${TAG}_pmc_calibration.json" <<'PY'
import csv, glob, json, sys
out, dst = sys.argv[1], sys.argv[2]
w, h, frames = 1232, 368, 256
npts = 1228 * 364
chunks = (npts - 4096) // 4096
pts = frames * chunks * 4096
res = {}
for m in (0, 30):
    vals = []
    for f in glob.glob(f"{out}/m{m}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == "FETCH_SIZE" and ("k7<" in r.get("Kernel_Name", "") or "k<" in r.get("Kernel_Name", "")):
                vals.append(float(r["Counter_Value"]))
    # FETCH_SIZE is reported in KiB-like units of 1024 B?  rocprofv3 derives it as bytes / 1024: keep both
    res[m] = {"dispatches": len(vals), "mean_counter": sum(vals) / max(1, len(vals))}
tpl = pts * 16.0
img = frames * (chunks * 4096 / w + 4) * w * 4.0  # the rows the chunks land on, fetched once
for unit in (1.0, 1024.0):
    a = res[0]["mean_counter"] * unit / tpl
    if 0.2 < a < 2.5:
        res["unit_bytes"] = unit
        res["template_stream_tally"] = a
        res["tap_stream_tally"] = (res[30]["mean_counter"] * unit - a * tpl) / img
res["known_bytes"] = {"template": tpl, "image_rows": img}
json.dump(res, open(dst, "w"), indent=1)
print(json.dumps(res))
PY
rm -rf $OUT/m0 $OUT/m30
