cd $GRAFT_REPO_ROOT
for G in 64 16 8 1; do echo "FINISH_G=$G"; MTG_FINISH_G=$G bash scripts/r6_quick.sh human-indel lib; done
