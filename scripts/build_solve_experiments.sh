#!/bin/bash
# Developer experiment: variants of the solver kernels with one cost removed at a time (results are WRONG on purpose; only kernel times
# are read, by scripts/solve_probe.py --lib). Builds locityper_amd/exp/liblocityper_hip_exp<N>.so from a patched copy of lcty_solve.hip.
#   solve_init_kernel:  bit 1: no record stores   bit 2: no tweak draws (windows without the per-chain shift)   bit 4: no LDS depth atomics
#                       bit 8: no exchange (barriers / prefix) between the passes
#   greedy_loop_kernel: bit 16: table values computed, not gathered   bit 32: records from 1 024 slots (cache-resident)
#                       bit 64: no duplicate check of the sample
#   bit 128: the loop kernels return at once (initialisation experiments: the records may be incomplete)
set -e
cd "$(dirname "$0")/../locityper_amd/csrc"
mkdir -p ../exp
OBJS=$(ls *.o | grep -v lcty_solve.o)
for N in "$@"; do
  sed -e "s|                if (nw > 1) recs\[slot\] = rec;|                if (nw > 1 \&\& !($N \& 1)) recs[slot] = rec;|" \
      -e "s|    if (V.tweak) {\$|    if (V.tweak \&\& !($N \& 2)) {|" \
      -e "s|                    if (t == a0) {\$|                    if (t == a0 \&\& !($N \& 4)) {|" \
      -e "s|        block_prefix_multi<RPT>(nt_k, ex_k, lane, wave, wave_sums, slot_k, eix_k, \&chunk_nt, \&chunk_ex);|        if ($N \& 8) { for (uint32_t k = 0; k < RPT; k++) { slot_k[k] = 256 * k + tid; eix_k[k] = 0; } chunk_nt = 256 * RPT; chunk_ex = 0; } else block_prefix_multi<RPT>(nt_k, ex_k, lane, wave, wave_sums, slot_k, eix_k, \&chunk_nt, \&chunk_ex);|" \
      -e "s|            g.vnew\[i\] = V->lut\[row + min(d_new, last)\];|            g.vnew[i] = ($N \& 16) ? -0.01 * d_new : V->lut[row + min(d_new, last)];|" \
      -e "s|            g.vold\[i\] = V->lut\[row + min(d_old, last)\];|            g.vold[i] = ($N \& 16) ? -0.01 * d_old : V->lut[row + min(d_old, last)];|" \
      -e "s|            const ChainRec\* r = \&recs\[cand ? c.pick : 0u\];|            const ChainRec* r = \&recs[(cand ? c.pick : 0u) \& (($N \& 32) ? 1023u : 0xFFFFFFFFu)];|" \
      -e "s|            unsigned long long dup_rows = __ballot(dup \&\& cand);|            unsigned long long dup_rows = ($N \& 64) ? 0ull : __ballot(dup \&\& cand);|" \
      -e "s|    if (flagged != 0u) return;|    if (flagged != 0u \|\| ($N \& 128)) return;|" \
      lcty_solve.hip > /tmp/lcty_solve_exp$N.hip
  echo "exp$N: $(diff lcty_solve.hip /tmp/lcty_solve_exp$N.hip | grep -c '^>') lines patched"
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -c /tmp/lcty_solve_exp$N.hip -o /tmp/lcty_solve_exp$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/liblocityper_hip_exp$N.so $OBJS /tmp/lcty_solve_exp$N.o -L/opt/rocm/lib -lrccl -lz -ldl -Wl,-rpath,/opt/rocm/lib
  echo built exp$N
done
