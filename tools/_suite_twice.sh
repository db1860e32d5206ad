# the whole GPU suite twice (flake hunting), smoke once
cd $GRAFT_REPO_ROOT
for i in 1 2; do timeout 3000 python -m pytest tests -q -m gpu -p no:cacheprovider 2>&1 | tail -1; done
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
