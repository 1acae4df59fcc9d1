#!/bin/bash
# diagnostic: per-phase stamps of single sites of the bench's site set (-DMTG_STAMPS build, then the product build again)
cd $GRAFT_REPO_ROOT
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc EXTRA="-DMTG_STAMPS" 2>&1 | grep -E "error"
timeout 600 python3 scripts/diag_one_gap.py "$@" 2>&1 | grep -E "stamps|^site" | tail -8
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E "error"
