cd $GRAFT_REPO_ROOT
python profiles/r06_chain_graph/leak_check.py 600 2>&1 | tail -10
