#!/bin/bash
# diagnostic: per-phase stamps of the walks that cross two SNPs exactly k apart (general bubble code), -DMTG_STAMPS build, then the product build again
cd $GRAFT_REPO_ROOT
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc EXTRA="-DMTG_STAMPS" 2>&1 | grep -E "error"
timeout 600 python3 scripts/diag_kk_gaps.py 2>&1 | grep -E "stamps|gaps whose|one wave" | tail -6
make -C mindthegap_amd/csrc clean >/dev/null; make -C mindthegap_amd/csrc 2>&1 | grep -E "error"
