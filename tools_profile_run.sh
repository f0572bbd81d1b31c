#!/bin/bash
# Profiling passes of bench.py on the GPU box (run through gpurun): kernel trace + stats, then one rocprofv3 --pmc pass per
# counter group (counters are collected in runs of their own, with --kernel-trace only).  Output: gpurun_out/prof_<tag>/...
# usage: tools_profile_run.sh <tag> <groups: stats,hbm,sq,l2,ta> [bench args...]
set -u
tag=$1; groups=$2; shift 2
out=gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
# one chunk in flight, as in the frames bench.py takes its roofline from (--profile-frames): with two, a kernel's interval in the trace
# includes the time it shares the GPU with the other chunk's kernels (the timed frames of a default run do overlap them)
export RT_STREAMS=${RT_STREAMS:-1}
python3 -c "import json; from raytracing_folder_amd import buildinfo as b; json.dump({'kernel_source_sha16': b.kernel_source_sha16(), 'build_flags': b.build_flags()}, open('$out/build_id.json', 'w'))"
run() {  # name, extra rocprof args...
  local name=$1; shift
  timeout -k 10 400 rocprofv3 --kernel-trace "$@" --output-format csv -d $out/$name -o $name -- python3 bench.py --no-cpu-baseline --profile-frames 0 "${BENCH_ARGS[@]}" > $out/$name.log 2>&1
  echo "$name rc=$?"
}
BENCH_ARGS=(--steps 2 --warmup 1 "$@")
case ",$groups," in *,stats,*) run stats --stats ;; esac
BENCH_ARGS=(--steps 1 --warmup 0 "$@")
case ",$groups," in *,hbm,*) run pmc_fetch --pmc FETCH_SIZE; run pmc_write --pmc WRITE_SIZE ;; esac
case ",$groups," in *,sq,*)
  run pmc_sq1 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY
  run pmc_sq2 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD
  run pmc_sq3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES ;; esac
case ",$groups," in *,ta,*)
  run pmc_ta1 --pmc TA_BUSY_avr TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TCP_GATE_EN1_sum
  run pmc_ta2 --pmc TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum ;; esac
case ",$groups," in *,l2,*) run pmc_l2 --pmc TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum ;; esac
find $out -name "*.csv" | head -30
