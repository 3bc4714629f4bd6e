#!/bin/bash
# registers / scratch / spill sites of the persistent PCG kernel (no GPU needed): compile solver.hip to gfx950 assembly
cd /root/repo/rootba_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -S --cuda-device-only solver.hip -o /tmp/solver.s 2>&1 | grep -E "error|warning: " | grep -v hip-link | head -20
for k in IfEE IdEE; do
  echo "k_pcgp<$k>:" $(grep -A60 "amdhsa_kernel _ZN3rba6k_pcgp$k" /tmp/solver.s | grep -E "private_segment_fixed|next_free_vgpr|next_free_sgpr" | tr -d '\t' | tr '\n' ' ')
  awk "/^_ZN3rba6k_pcgp${k}vNS_8PgParamsIT_EE:/,/\.Lfunc_end/" /tmp/solver.s > /tmp/pcgp_$k.s
  echo "  scratch ops: $(grep -c scratch_ /tmp/pcgp_$k.s)  lines: $(wc -l < /tmp/pcgp_$k.s)"
done
