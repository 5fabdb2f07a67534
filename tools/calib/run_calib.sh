#!/bin/bash
# run_calib.sh -- on the GPU box: FETCH_SIZE and WRITE_SIZE of tools/calib/fetch_calib (known byte counts per access shape), two --pmc passes
set -u
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out; mkdir -p $OUT
[ -x $R/tools/calib/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o $R/tools/calib/fetch_calib $R/tools/calib/fetch_calib.hip
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/calib_fetch -o p -- $R/tools/calib/fetch_calib > $OUT/calib_run.json 2> $OUT/calib_fetch.err
timeout 120 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/calib_write -o p -- $R/tools/calib/fetch_calib > /dev/null 2> $OUT/calib_write.err
F=$(find $OUT/calib_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/calib_write -name "*counter_collection.csv" | head -1)
python $R/tools/calib/summarize_calib.py "$F" "$W" > $OUT/calibration.json
cat $OUT/calibration.json
