# GPU box: rocprofv3 kernel-trace summary of the MLP-root training step (bench.py --root <net>).  usage: prof_mlp.sh TAG NET
TAG=${1:-mlp}; NET=${2:-mlp2x16}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_${TAG} -o p -- python bench.py --root $NET --steps 40 --warmup 40 > gpurun_out/prof_${TAG}.log 2>&1
tail -1 gpurun_out/prof_${TAG}.log | cut -c1-400
python - <<PY
import glob,csv
f=glob.glob("gpurun_out/prof_${TAG}/**/*kernel_stats.csv",recursive=True)[0]
for r in list(csv.DictReader(open(f)))[:14]: print(r["Name"][:90], r["Calls"], r["AverageNs"], r["Percentage"])
PY
