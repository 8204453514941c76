#!/bin/bash
# round 3, GPU session 1: tests, all-shard rehearsal (wave scatter off / on), default bench, self-launched 2-rank rehearsal
set -o pipefail
R=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$R/gpurun_out/r03; mkdir -p $OUT; cd $R
timeout -k 10 400 python -m pytest tests -m gpu -x -q > $OUT/pytest.log 2>&1 || { tail -30 $OUT/pytest.log; exit 1; }
tail -2 $OUT/pytest.log
echo "== B scatter off"; HIPRZ_WAVE_SCATTER=0 timeout -k 10 200 python tools/shard_scaling.py --config B --shards 1,8 > $OUT/shards_B_scatter0.jsonl 2> $OUT/shards.err || { tail -5 $OUT/shards.err; exit 1; }
cat $OUT/shards_B_scatter0.jsonl | cut -c 1-420
echo "== B scatter auto"; timeout -k 10 200 python tools/shard_scaling.py --config B --shards 1,2,4,8 > $OUT/shards_B.jsonl 2>> $OUT/shards.err || { tail -5 $OUT/shards.err; exit 1; }
cat $OUT/shards_B.jsonl | cut -c 1-420
echo "== B scatter always"; HIPRZ_WAVE_SCATTER=1 timeout -k 10 200 python tools/shard_scaling.py --config B --shards 1 > $OUT/shards_B_scatter1.jsonl 2>> $OUT/shards.err || { tail -5 $OUT/shards.err; exit 1; }
cat $OUT/shards_B_scatter1.jsonl | cut -c 1-300
echo "== D, E"; timeout -k 10 500 python tools/shard_scaling.py --config D,E --shards 1,8 --steps 10 > $OUT/shards_DE.jsonl 2>> $OUT/shards.err || { tail -5 $OUT/shards.err; exit 1; }
cat $OUT/shards_DE.jsonl | cut -c 1-420
echo "== bench default"; timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $OUT/bench_B_s1.json 2> $OUT/bench.err || { tail -5 $OUT/bench.err; exit 1; }
cut -c 1-300 $OUT/bench_B_s1.json
echo "== self-launched 2 ranks on one GPU"; timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 2 --rehearse-on-one-gpu --verify-gather --min-seconds 1 > $OUT/rehearse2.json 2> $OUT/rehearse2.err || { tail -20 $OUT/rehearse2.err; exit 1; }
cut -c 1-300 $OUT/rehearse2.json
