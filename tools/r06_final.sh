#!/bin/bash
# final pass of the round: the whole GPU suite, the profiles (tools/r06_profiles.sh), then more iterations of the C++ RCCL soak
root=${GRAFT_REPO_ROOT:-$PWD}
cd $root
bash tools/r06_suite.sh || exit 1
bash tools/r06_profiles.sh || exit 1
[ -x tools/soak/mc_soak ] && bash tools/soak/run_soak.sh ${1:-420} 5 5
