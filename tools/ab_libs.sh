# usage: bash tools/ab_libs.sh <tag> <workload> name1 name2 ...   same-box A/B of library builds build_ab/libgsrast_<name>.so (GSRAST_LIB): fps and the blend kernels, two rounds interleaved
tag=$1; wl=$2; shift 2
mkdir -p gpurun_out/$tag
for r in 1 2; do for v in "$@"; do GSRAST_LIB=$PWD/build_ab/libgsrast_$v.so python bench.py --no-cpu-baseline --breakdown-steps 50 --steps 200 --workload $wl > gpurun_out/$tag/${wl}_${v}_$r.json 2>/dev/null; done; done
python - "$tag" <<'PY'
import json,glob,sys
for f in sorted(glob.glob("gpurun_out/%s/*.json" % sys.argv[1])):
    d=json.load(open(f)); k=d.get("kernels_ms_per_view") or d["kernels_ms_per_step"]; print(f.split("/")[-1], d["value"], k.get("k_blend_fwd"), k.get("k_blend_bwd_tile"), k.get("k_sum_rows"), k.get("k_project"))
PY
