#!/bin/bash
# the non-cluster lines of the configs[4] report on the tenth set (what the test's sha256 covers beside the 14.6 M cluster lines)
set -e
D=$(mktemp -d); cd $D
$GRAFT_REPO_ROOT/build/gen_fqb -v 2 -P 30000000 -C 160000 -G 300000000 -e 0.0005 -s 3 -o g3t.fqb -fa g3t 2>/dev/null
$GRAFT_REPO_ROOT/bin/hash10x-amd -B 27 --readFQB g3t.fqb --hashDepthRange 6 45 --cluster 1 0 -o g3t.report.txt --cribBuild g3t.A.fa g3t.B.fa --clusterReport 1 0 --clusterSplit --cribSummary -o - --writeHash g3t.split.hash > stdout.txt 2> stderr.txt
grep -v "CODE_CLUSTER\|CLUSTER_SUMMARY" g3t.report.txt > $GRAFT_REPO_ROOT/gpurun_out/r5_c5_other.txt || true
head -c 3000 g3t.report.txt > $GRAFT_REPO_ROOT/gpurun_out/r5_c5_head.txt
tail -c 2000 g3t.report.txt > $GRAFT_REPO_ROOT/gpurun_out/r5_c5_tail.txt
cp stdout.txt $GRAFT_REPO_ROOT/gpurun_out/r5_c5_stdout.txt; tail -c 3000 stderr.txt > $GRAFT_REPO_ROOT/gpurun_out/r5_c5_stderr.txt
ls -la >> $GRAFT_REPO_ROOT/gpurun_out/r5_c5_stdout.txt
cd /; rm -rf $D
