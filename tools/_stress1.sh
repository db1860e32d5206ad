# round 6: first contact of tools/stream_stress.py -- which variant diverges, under which load
cd $GRAFT_REPO_ROOT
for load in busy idle none; do
  echo "== sharded, load=$load"; timeout 900 python tools/stream_stress.py --model sharded --variants one,two,two_eager --trials 24 --load $load 2>&1 | grep RESULT
done
echo "== unsharded, load=busy"; timeout 900 python tools/stream_stress.py --model unsharded --variants one,two,two_eager --trials 24 --load busy 2>&1 | grep RESULT
