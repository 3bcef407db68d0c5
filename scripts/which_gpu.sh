#!/bin/bash
# Which KFD topology node is the GPU this container can see (the number in "Memory access fault by GPU node-N")
for n in /sys/class/kfd/kfd/topology/nodes/*; do
  if cat $n/name >/dev/null 2>&1; then
    nm=$(cat $n/name); sz=$(grep -c . $n/properties 2>/dev/null)
    if [ -n "$nm" ]; then echo "visible kfd node $(basename $n): name=$nm simd_count=$(grep simd_count $n/properties | cut -d' ' -f2) host=$(hostname)"; fi
  fi
done
