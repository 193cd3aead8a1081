set -e
cd $GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
for w in 2 3 4 5; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DGTO_OBS_MIN_WAVES=$w gto_api.hip -o /tmp/lib_w$w.so 2>/dev/null; done
cd $GRAFT_REPO_ROOT
for w in 2 3 4 5; do echo "== min waves $w"; GTO_HIP_LIB=/tmp/lib_w$w.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
