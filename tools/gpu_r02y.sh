cd $GRAFT_REPO_ROOT
python -c "import __graft_entry__ as g; g.build()" > /dev/null 2>&1 || exit 1
bash tools/pmc_wino.sh r02y tower 2>&1 | grep -v amdgpu.ids | tail -12
