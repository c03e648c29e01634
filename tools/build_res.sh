#!/bin/bash
# build the product library with the kernel resource remarks; print those of the C2 extract kernel (development aid)
cd /root/repo/kmernator_amd/csrc || exit 1
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -munsafe-fp-atomics -w -Rpass-analysis=kernel-resource-usage -shared -o libkmernator_amd.so kmr_api.hip > /tmp/build_res.log 2>&1
grep -m3 "error" -A5 /tmp/build_res.log
pat=${1:-_ZN3kmr14extract_kernelILi1ELb0ENS_8LinearOpILi1ELb0EEELb0}
grep -A9 "Function Name: $pat" /tmp/build_res.log | grep -E "Name|SGPRs|VGPRs|Scratch" | sed 's/.*remark: *//'
