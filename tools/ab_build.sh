#!/bin/bash
# A/B libraries with the product's own flags (__graft_entry__.HIPCC_FLAGS): tools/ab_build.sh <tag> [-DNAME=v ...] -> /tmp/lib_<tag>.so
tag=$1; shift
cd ${GRAFT_REPO_ROOT:-/root/repo}/grasptrajopt_amd/csrc
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -munsafe-fp-atomics -mllvm -amdgpu-kernarg-preload-count=16 "$@" gto_api.hip -o /tmp/lib_$tag.so -Rpass-analysis=kernel-resource-usage 2> /tmp/lib_$tag.remarks || { tail -5 /tmp/lib_$tag.remarks; exit 1; }
python3 - "$tag" <<'P'
import re,sys
rows=[];
for line in open(f"/tmp/lib_{sys.argv[1]}.remarks"):
    m=re.search(r"Function Name: (\S+)",line)
    if m: rows.append({"name":m.group(1)}); continue
    m=re.search(r"remark:\s+(VGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]): (\d+)",line)
    if m and rows: rows[-1][m.group(1).split(" ")[0]]=int(m.group(2))
for r in rows:
    if any(k in r["name"] for k in ("k_obstacle_gramILi8ELi1ELb0ELb1","k_obstacle_gramILi16ELi1ELb0ELb1","k_lm_stepILi4","k_lm_stepILi8","k_lm_step_wide")):
        print(" ", r)
P
