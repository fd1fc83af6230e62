# build the engine of another commit into build_ab/libA.so (baseline of tools/ab_lib.sh):  bash tools/build_ref_lib.sh <commit> [name]
set -e
C=${1:-HEAD}; N=${2:-libA}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/svln_ref.XXXX)
git -C $ROOT worktree add --detach $W $C > /dev/null
bash $W/streamvln_amd/csrc/build.sh > /dev/null
mkdir -p $ROOT/build_ab
cp $W/streamvln_amd/libstreamvln_hip.so $ROOT/build_ab/$N.so
git -C $ROOT worktree remove --force $W
echo "built build_ab/$N.so from $C"
