#!/bin/bash
# r05_final_collect.sh -- on the GPU box: everything profiles/r05_final_* and the round's latency records are made of, in one go (about nine minutes).
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
timeout 900 bash $R/tools/collect_profiles.sh r05_final > $OUT/r05_final_collect.log 2>&1
cd /tmp && export TMPDIR=/tmp
# a lone capture's kernels (fsample-6, the library's automatic choice: k_scan, k_pipe, k_win_redo_finish)
rocprofv3 --kernel-trace --output-format rocpd -d $OUT/r05_single -o p -- python $R/tools/r05_window_latency.py --profile-single 30 > $OUT/r05_single.log 2>&1
DB=$(find $OUT/r05_single -name "*_results.db" | head -1)
python $R/tools/rocprof_kernels.py "$DB" --last 3 > $OUT/r05_single_capture_kernels.json 2>> $OUT/r05_single.log
rm -rf $OUT/r05_single
cd $R
timeout 200 python tools/pipe_timeline.py > $OUT/r05_pipe_timeline.txt 2>&1
timeout 300 python tools/r05_window_latency.py --reps 40 > $OUT/r05_window_latency.json 2> $OUT/r05_window_latency.err
timeout 900 python bench.py > $OUT/r05_final_bench.json 2> $OUT/r05_final_bench.err
tail -c 400 $OUT/r05_final_bench.json; tail -3 $OUT/r05_final_bench.err; cat $OUT/r05_pipe_timeline.txt | tail -14; head -30 $OUT/r05_single_capture_kernels.json
