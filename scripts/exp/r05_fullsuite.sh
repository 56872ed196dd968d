( time python -m pytest tests/ -x -q -m gpu --durations=25 ) > gpurun_out/r05_fullsuite.log 2>&1
tail -45 gpurun_out/r05_fullsuite.log
