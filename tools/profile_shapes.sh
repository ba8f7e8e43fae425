#!/bin/bash
# rocprofv3 --kernel-trace --stats for every bench workload (kernel time per shape), run through gpurun from the repo root:
#   tools/profile_shapes.sh <tag> ["<workload[:streams]> ..."]   -> gpurun_out/shapes_<tag>.txt
set -u
TAG=${1:-r1}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out/shapes_$TAG
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
SUM=$ROOT/gpurun_out/shapes_$TAG.txt
echo "# rocprofv3 --kernel-trace --stats, bench.py --no-cpu-baseline --steps 200 --warmup 20 --workload <w> (256 streams, 1 hop per launch)" > "$SUM"
# "<workload>[:streams]" (default 256 streams); the :1024 lines run the low-LDS companions / two-workgroups-per-CU builds
LIST=${2:-"fe_t fe_b fe_s fe_m fe_l fe48_t fe48_b fe48_b_h480 fe48_s fe48_m fe48_l fe_tk_b fe_dprnn_t fe_dprnn_b fe_dprnn_s fe_dprnn_m fe_dprnn_l fe_dpt_t fe_dpt_b fe_dpt_s fe_dpt_m fe_ln_b bsrnn_xxt bsrnn_xt bsrnn_t bsrnn_s fspen lisennet \
          fe_t:1024 fe_b:1024 fe_s:512 fe48_t:1024 fe48_b:1024 fe48_b_h480:512 fe_dprnn_t:1024 fe_dpt_t:1024 fe_ln_b:1024 bsrnn_xxt:1024 bsrnn_xt:1024 fspen:4096 lisennet:4096"}
for ws in $LIST; do
  w=${ws%%:*}; st=256; [[ $ws == *:* ]] && st=${ws##*:}
  rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/$ws" -o s -- python $ROOT/bench.py --no-cpu-baseline --steps 200 --warmup 20 --workload $w --streams $st > "$OUT/$ws.json" 2> "$OUT/$ws.err"
  python - "$OUT/$ws" "$ws" "$OUT/$ws.json" >> "$SUM" <<'PY'
import csv, glob, json, sys
d, w, j = sys.argv[1:4]
line = open(j).read().strip().splitlines()[-1]
b = json.loads(line)
for f in glob.glob(d + "/**/*kernel_stats.csv", recursive=True):
    for row in csv.DictReader(open(f)):
        if "frame_kernel" in row["Name"]:
            print(f"{w:16s} {row['Name'][:70]:70s} calls={row['Calls']} avg_ns={float(row['AverageNs']):.0f}  bench under tracer: {b['value']:.0f} frames/s, frac {b['roofline']['frac']:.4f}")
PY
done
cat "$SUM"
