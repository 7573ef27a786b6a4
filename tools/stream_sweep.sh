cd $GRAFT_REPO_ROOT
run() { echo "== $*"; env "$@" MODES=1 timeout 120 python tools/stream_probe.py 2>&1 | grep "FACPPG_STREAM="; }
timeout 300 python -m pytest tests/test_gpu_stream.py -x -q 2>&1 | tail -3
run A=1
run FACPPG_STREAM_SPARE_CUS=-1
run FACPPG_STREAM_SPARE_CUS=4
run FACPPG_STREAM_SPARE_CUS=16
run FACPPG_STREAM_CHUNK=32
run FACPPG_STREAM_GROUPS=1
