set -e
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
W=/tmp/lhw; rm -rf $W; mkdir -p $W/sim
NEWICK="(((chimp:6,human:6):81,(mouse:17,rat:17):70):6,dog:93)"
cat > $W/gen.sh <<EOS
seed 10
load -i $GRAFT_REPO_ROOT/tests/golden/example_data.tab -t 1
tree $NEWICK
lambda -s
genfamily $W/sim/rnd -t 40
EOS
cafe_amd/bin/cafehip $W/gen.sh > /dev/null
for mode in "1 1" "2 1" "4 1" "8 1" "2 0" "4 0"; do set -- $mode
cat > $W/lh.sh <<EOS
seed 10
load -i $GRAFT_REPO_ROOT/tests/golden/example_data.tab -t 1
tree $NEWICK
lambda -s
lhtest -d $W/sim -t (((1,1)1,(2,2)2)2,2) -l 0.0107527 -o $W/lh_$1_$2.out
EOS
if [ $1 = 1 ]; then A=""; else A="--gpus $1 --same-device"; fi
CAFEHOST_TIMING=1 CAFEHOST_LHTEST_DEAL=$2 cafe_amd/bin/cafehip $A $W/lh.sh 2>&1 >/dev/null | grep "lhtest:" | sed "s/^/ranks=$1 deal=$2: /"
done
md5sum $W/lh_*.out
