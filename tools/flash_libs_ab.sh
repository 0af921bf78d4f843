# per-kernel durations of the attention kernels for several library builds (IX_LIB_PATH).  usage: sh tools/flash_libs_ab.sh "<shapes>" outdir lib1.so lib2.so ...
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPES=$1; OUT=$2; shift; shift
mkdir -p $OUT
for S in $SHAPES; do
for L in "$@"; do
  rm -rf $OUT/p; mkdir -p $OUT/p
  IX_LIB_PATH=$GRAFT_REPO_ROOT/interactron_amd/lib/$L rocprofv3 --kernel-trace --stats -d $OUT/p -o p --output-format csv -- python3 tools/flash_bench.py $S > $OUT/out_${S}_$L.txt 2>&1
  echo "== $S $L"; grep "flash:" $OUT/out_${S}_$L.txt | cut -c1-150
  python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/p/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'flash' in r['Name']: print("   %-62s calls %4s avg %9.1f us"%(r['Name'][:62],r['Calls'],float(r['AverageNs'])/1e3))
PY
done
done
rm -rf $OUT/p
