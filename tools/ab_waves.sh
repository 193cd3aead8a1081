set -e
cd $GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
for cfg in "4 80" "5 80" "6 72" "7 72"; do set -- $cfg; hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DGTO_OBS_MIN_WAVES=$1 -DGTO_LIST_CAP=$2 gto_api.hip -o /tmp/lib_w$1.so 2>/dev/null; done
cd $GRAFT_REPO_ROOT
for w in 4 5 6 7; do echo "== min waves $w"; GTO_HIP_LIB=/tmp/lib_w$w.so python bench.py --steps 8 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d['roofline']['avg_launch_us'], d['roofline']['frac'])"; done
