#!/bin/bash
# A/B: register budget of the base-placement kernel (waves per SIMD the allocator leaves room for)
cd $GRAFT_REPO_ROOT/grasptrajopt_amd/csrc
for w in 1 2 3 4; do hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -DGTO_BASE_MIN_WAVES=$w gto_api.hip -o /tmp/lib_b$w.so -Rpass-analysis=kernel-resource-usage 2>&1 | grep -A8 "k_base_solve" | grep -i "VGPRs:\|Scratch\|Occupancy" | tr '\n' ' '; echo " <- min waves $w"; done
cd $GRAFT_REPO_ROOT
for w in 1 2 3 4; do echo "min waves $w:"; GTO_HIP_LIB=/tmp/lib_b$w.so python tools/base_rate.py --sets 64 2>/dev/null | grep effort; GTO_HIP_LIB=/tmp/lib_b$w.so python tools/base_rate.py --sets 1024 2>/dev/null | grep effort; done
