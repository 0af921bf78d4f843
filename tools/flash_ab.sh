# A/B of two library builds on the flash kernels: per-kernel average durations from rocprofv3 (tools/flash_bench.py shapes)
# usage: sh tools/flash_ab.sh "<shapes>" libA.so libB.so
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPES=${1:-fusion}
shift
for L in "$@"; do
  rm -rf gpurun_out/fab; mkdir -p gpurun_out/fab
  IX_LIB_PATH=$GRAFT_REPO_ROOT/interactron_amd/lib/$L rocprofv3 --kernel-trace --stats -d gpurun_out/fab -o p --output-format csv -- python3 tools/flash_bench.py $SHAPES > gpurun_out/fab/out.txt 2>&1
  echo "== $L"; grep "flash:" gpurun_out/fab/out.txt | cut -c1-150
  python3 - <<'PY'
import csv,glob
f=glob.glob('gpurun_out/fab/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'flash_' in r['Name']: print("   %-60s calls %4s avg %9.1f us"%(r['Name'][:60],r['Calls'],float(r['AverageNs'])/1e3))
PY
done
