cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout 200 python tests/stage_profile.py 2>&1 | tail -8
GATRES_FUSED_SPLIT=7 timeout 200 python tests/stage_profile.py 2>&1 | tail -5
