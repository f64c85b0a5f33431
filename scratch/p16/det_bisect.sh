#!/bin/bash
cd $GRAFT_REPO_ROOT
for e in "$@"; do
  echo "==== $e"; env MONOCON_HIP_SIDE_SYNC=$e python scratch/p16/dbg_pgrad_det.py f16x2p 2 2>&1 | grep "differ" | head -3
done
