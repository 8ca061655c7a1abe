#!/bin/bash
# tools/box_info.sh : what distinguishes one GPU box from another (clocks, partition / XNACK mode, driver parameters)
echo "== xnack / noretry"; /opt/rocm/bin/rocminfo 2>/dev/null | grep -iE "xnack|Marketing|Compute Unit|Max Clock|Name:  *gfx" | sort | uniq -c | head -12
cat /sys/module/amdgpu/parameters/noretry 2>/dev/null; echo "HSA_XNACK=${HSA_XNACK:-unset}"
echo "== rocm-smi"; /opt/rocm/bin/rocm-smi --showcomputepartition --showmemorypartition --showclocks --showperflevel --showpower --showmaxpower 2>&1 | grep -vE "^=|^$" | head -40
echo "== kernel params"; for f in /sys/module/amdgpu/parameters/{vm_fragment_size,sched_policy,mes,ras_enable,ppfeaturemask}; do echo "$f $(cat $f 2>/dev/null)"; done
uname -r; cat /proc/cmdline 2>/dev/null | tr ' ' '\n' | grep -iE "iommu|amd|numa" | head
