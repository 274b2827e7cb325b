#!/bin/bash
# rocprofv3 evidence for one bench.py configuration (GPU box): kernel trace + stats, and (PMC=1) three separate counter passes
# (MFMA busy, FETCH_SIZE, WRITE_SIZE -- never combined with a trace, see the gpurun rules), summarised into
# gpurun_out/prof_<tag>/{kernel_stats.md,pmc.md}.   scripts/profile_job.sh <tag> <PMC 0|1> <bench.py args ...>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}
TAG=$1; PMC=$2; shift 2
O=$R/gpurun_out/prof_$TAG
rm -rf $O; mkdir -p $O
export TMPDIR=/tmp
cd /tmp
timeout 400 rocprofv3 --kernel-trace --stats -d $O/kt -o kt -- python $R/bench.py --steps 8 --warmup 3 --no-cpu-baseline --no-extra-legs "$@" > $O/kt.log 2>&1
grep "^{\"metric\"" $O/kt.log | tail -1 | cut -c1-400 > $O/bench_line_profiled.json
DB=$(find $O/kt -name "*.db" | head -1)
python $R/scripts/summarize_profile.py db $DB > $O/kernel_stats.md 2>$O/sum.err
if [ "$PMC" = "1" ]; then
  for P in "SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" "FETCH_SIZE" "WRITE_SIZE"; do
    N=$(echo $P | cut -d' ' -f1)
    timeout 400 rocprofv3 --pmc $P --output-format csv -d $O/pmc_$N -o pmc -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extra-legs --no-overlap "$@" > $O/pmc_$N.log 2>&1
  done
  python $R/scripts/summarize_profile.py pmc $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE > $O/pmc.md 2>>$O/sum.err
  rm -rf $O/pmc_SQ_VALU_MFMA_BUSY_CYCLES $O/pmc_FETCH_SIZE $O/pmc_WRITE_SIZE
fi
rm -rf $O/kt
echo "== $TAG"; head -9 $O/kernel_stats.md; [ "$PMC" = "1" ] && head -6 $O/pmc.md
cd $R
