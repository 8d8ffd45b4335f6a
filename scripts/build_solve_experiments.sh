#!/bin/bash
# Developer experiment: variants of the greedy loop with one cost removed at a time (results are WRONG on purpose; only the time per
# iteration is read). Builds locityper_amd/exp/liblocityper_hip_exp<N>.so from a patched copy of lcty_solve.hip.
#   bit 1: window weights = 1.0 (no weight gather)   bit 2: table values computed, not gathered   bit 4: records from 1 024 slots (cache-resident)
set -e
cd "$(dirname "$0")/../locityper_amd/csrc"
mkdir -p ../exp
for N in "$@"; do
  sed -e "s|weight\[i\] = ww\[w\[i\]\];|weight[i] = ($N \& 1) ? 1.0 : ww[w[i]];|" \
      -e "s|vnew\[i\] = V->lut\[row + min(d_new, last)\];|vnew[i] = ($N \& 2) ? -0.01 * d_new : V->lut[row + min(d_new, last)];|" \
      -e "s|vold\[i\] = V->lut\[row + min(d_old, last)\];|vold[i] = ($N \& 2) ? -0.01 * d_old : V->lut[row + min(d_old, last)];|" \
      -e "s|const uint32_t slot = cand ? s.pick : 0u;|const uint32_t slot = (cand ? s.pick : 0u) \& (($N \& 4) ? 1023u : 0xFFFFFFFFu);|" \
      lcty_solve.hip > /tmp/lcty_solve_exp$N.hip
  diff <(grep -c . lcty_solve.hip) <(grep -c . /tmp/lcty_solve_exp$N.hip)
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I. -c /tmp/lcty_solve_exp$N.hip -o /tmp/lcty_solve_exp$N.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../exp/liblocityper_hip_exp$N.so lcty_api.o lcty_locus.o lcty_reads.o lcty_score.o lcty_prefilter.o /tmp/lcty_solve_exp$N.o lcty_transfer.o lcty_recruit.o lcty_comm.o lcty_io.o lcty_bam.o -L/opt/rocm/lib -lrccl -lz -ldl -Wl,-rpath,/opt/rocm/lib
  echo built exp$N
done
