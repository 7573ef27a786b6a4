cd $GRAFT_REPO_ROOT
for T in 64 100 130 150 180 230 250 300 400 600 1000; do
  echo "== T=$T"; T=$T CONFIGS="FACPPG_STREAM=0;A=1" REPS=2 N=6 timeout 200 python tools/stream_probe.py 2>&1 | grep median | cut -c1-120
done
