#!/bin/bash
# rocprofv3 of bench.py for every BASELINE config: kernel trace and counter passes (each in a run of its own), all
# with the batch as one launch sequence on one stream (--shards 1: kernels of parts on several streams overlap and
# their durations stop adding up); one more kernel trace of config 3 over 400 steps, where bench.py steps the batch
# as three parts on three streams.  PROFILE_GIT_HEAD (the commit of the tree, which the GPU box cannot know) goes
# into the summaries.  They land under gpurun_out/profiles_<tag>/; copy what is to be kept into profiles/.
set -o pipefail
tag=${1:-r02}
configs=${2:-"3 2 4 5"}        # e.g. tools/profile_all.sh r02f 5
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
out=gpurun_out/profiles_$tag
mkdir -p $out
db() { ls -t $1/*/*.db $1/*.db 2>/dev/null | head -1; }
for c in $configs; do
  case $c in 2) A="--steps 300 --warmup 50";; 3) A="--steps 100 --warmup 20";; 4) A="--steps 10 --warmup 2";; 5) A="--steps 64 --warmup 64";; esac
  B="python3 bench.py --config $c $A --no-cpu-baseline --shards 1"
  timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_c$c -o r -- $B > $out/c${c}_trace_bench.json 2> $out/c${c}_trace.err || { echo "trace c$c failed"; tail -3 $out/c${c}_trace.err; exit 1; }
  python3 tools/db_summary.py stats $out/${tag}_c${c}_kernel_stats.csv $(db gpurun_out/prof_${tag}_c$c) > /dev/null || exit 1
  if [ $c = 3 ]; then
    timeout -k 10 300 rocprofv3 --kernel-trace --stats -d gpurun_out/prof_${tag}_c3s -o r -- python3 bench.py --config 3 --steps 400 --warmup 20 --no-cpu-baseline > $out/c3_streams_trace_bench.json 2> $out/c3_streams_trace.err || { echo "trace c3 streams failed"; exit 1; }
    python3 tools/db_summary.py stats $out/${tag}_c3_streams_kernel_stats.csv $(db gpurun_out/prof_${tag}_c3s) > /dev/null || exit 1
    rm -rf gpurun_out/prof_${tag}_c3s
  fi
  PMC=""
  if true; then
    case $c in 2) P="python3 bench.py --config 2 --steps 40 --warmup 10 --no-cpu-baseline --shards 1";;
               3) P="python3 bench.py --config 3 --steps 20 --warmup 5 --no-cpu-baseline --shards 1";;
               4) P="python3 bench.py --config 4 --steps 4 --warmup 1 --no-cpu-baseline --shards 1";;
               5) P="python3 bench.py --config 5 --steps 64 --warmup 64 --no-cpu-baseline";; esac
    timeout -s KILL 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES SQ_WAIT_INST_ANY -d gpurun_out/pmc_${tag}_c${c}_sq -- $P > $out/c${c}_pmc_sq.log 2>&1 || { echo "pmc sq c$c failed"; exit 1; }
    timeout -s KILL 200 rocprofv3 --pmc FETCH_SIZE -d gpurun_out/pmc_${tag}_c${c}_fetch -- $P > $out/c${c}_pmc_fetch.log 2>&1 || { echo "pmc fetch c$c failed"; exit 1; }
    timeout -s KILL 200 rocprofv3 --pmc WRITE_SIZE -d gpurun_out/pmc_${tag}_c${c}_write -- $P > $out/c${c}_pmc_write.log 2>&1 || { echo "pmc write c$c failed"; exit 1; }
    python3 tools/db_summary.py pmc $out/${tag}_c${c}_pmc.json $(db gpurun_out/pmc_${tag}_c${c}_sq) $(db gpurun_out/pmc_${tag}_c${c}_fetch) $(db gpurun_out/pmc_${tag}_c${c}_write) > /dev/null || exit 1
    PMC=$out/${tag}_c${c}_pmc.json
  fi
  PROFILE_COMMAND="$B" python3 tools/db_summary.py summary $out/summary_c$c.json $out/${tag}_c${c}_kernel_stats.csv $PMC > /dev/null || exit 1
  rm -rf gpurun_out/prof_${tag}_c$c gpurun_out/pmc_${tag}_c${c}_sq gpurun_out/pmc_${tag}_c${c}_fetch gpurun_out/pmc_${tag}_c${c}_write
  echo "config $c:"; cat $out/${tag}_c${c}_kernel_stats.csv | cut -c1-150
done
