#!/bin/bash
# config 5's sampler in fresh processes for several staggers of the chain-state arrays (GLMMR_MCML_CM_STAGGER): chain-steps/s
cd "$GRAFT_REPO_ROOT"
for s in 0 4096 65536 1048576 69888 0 4096 65536 1048576 69888 0 4096 65536 1048576 69888; do
  printf "stagger %8d: " $s
  GLMMR_MCML_CM_STAGGER=$s python3 scripts/time_cfg.py cfg5 1024 2>/dev/null | grep -E "hmc warm=50|fwd avg" | tr "\n" " " | sed -e 's/leapfrog [0-9]* //' -e 's/acc=.*fwd avg/ fwd avg/'
  echo
done
