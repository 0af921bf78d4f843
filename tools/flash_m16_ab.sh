# A/B of the two head-dim-64 kernel families (csrc/flash16.hip vs csrc/flash.hip) on the flash kernels of the step's shapes:
# per-kernel average durations from rocprofv3, one run per shape and family.  usage: sh tools/flash_m16_ab.sh "<shapes>" <outdir>
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
SHAPES=${1:-fusion}
OUT=${2:-gpurun_out/fab16}
mkdir -p $OUT
for S in $SHAPES; do
for M in 1 0; do
  rm -rf $OUT/p; mkdir -p $OUT/p
  IX_FLASH_M16=$M rocprofv3 --kernel-trace --stats -d $OUT/p -o p --output-format csv -- python3 tools/flash_bench.py $S > $OUT/out_${S}_m16_$M.txt 2>&1
  echo "== $S IX_FLASH_M16=$M"; grep "flash:" $OUT/out_${S}_m16_$M.txt | cut -c1-150
  python3 - $OUT <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+'/p/**/p_kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'flash' in r['Name'] or 'attn_split' in r['Name']: print("   %-62s calls %4s avg %9.1f us"%(r['Name'][:62],r['Calls'],float(r['AverageNs'])/1e3))
PY
done
done
rm -rf $OUT/p
