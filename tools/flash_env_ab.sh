# per-kernel durations of the attention kernels under two settings of one environment variable (same library).
# usage: sh tools/flash_env_ab.sh "<shapes>" outdir VAR val1 val2 [repeats]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPES=$1; OUT=$2; VAR=$3; V1=$4; V2=$5; REP=${6:-2}
mkdir -p $OUT
for S in $SHAPES; do
for R in $(seq $REP); do
for V in $V1 $V2; do
  rm -rf $OUT/p; mkdir -p $OUT/p
  export $VAR=$V
  rocprofv3 --kernel-trace --stats -d $OUT/p -o p --output-format csv -- python3 tools/flash_bench.py $S > $OUT/out_${S}_${VAR}_$V.txt 2>&1
  echo "== $S $VAR=$V"; grep "flash:" $OUT/out_${S}_${VAR}_$V.txt | cut -c1-150
  python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/p/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'flash' in r['Name']: print("   %-62s calls %4s avg %9.1f us"%(r['Name'][:62],r['Calls'],float(r['AverageNs'])/1e3))
PY
done
done
done
rm -rf $OUT/p
