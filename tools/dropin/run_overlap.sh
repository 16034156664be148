#!/bin/bash
# rocprofv3 kernel trace of the harness + overlap.py.   usage: run_overlap.sh <tag> <threads> <blocks> <run_ahead>
tag=$1; nt=$2; nb=$3; ra=$4
conf=$(python3 tools/dropin/make_conf.py /tmp/folve_dropin_conf)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/tr_$tag
rocprofv3 --kernel-trace --output-format csv -d /tmp/tr_$tag -- $GRAFT_REPO_ROOT/tools/dropin/dropin_threads "$conf" $nt $nb 1 run_ahead=$ra ${@:5} > /tmp/tr_$tag.log 2>&1
cd $GRAFT_REPO_ROOT
{ grep "^threads" /tmp/tr_$tag.log; python3 tools/dropin/overlap.py /tmp/tr_$tag $TIMELINE; } > gpurun_out/overlap_$tag.txt 2>&1
cat gpurun_out/overlap_$tag.txt
