# Round 5, first GPU call: the tree on config 4 and on the headline (new line: fold off in the timed region, single_job, roofline
# from the job alone), and the three candidates of tools/proto/ beside it as variant builds, each with its parity subset.
# Before the call, here:  tools/make_prof.sh; tools/make_variant.sh emit tools/proto/lz77_vector_emitter.patch;
#   tools/make_variant.sh emit_prof -DZPQ_LZ_PROFILE tools/proto/lz77_vector_emitter.patch;
#   tools/make_variant.sh pack tools/proto/lz77_pack_literals_in_token_kernel.patch;
#   tools/make_variant.sh maskspec tools/proto/lz77_spec_kernel_mask_evaluator.patch
cd $GRAFT_REPO_ROOT
export PYTHONUNBUFFERED=1
T=${1:-r05a}
S0=$(date +%s)
line() { python -c "
import json,sys
d=json.loads(sys.stdin.read()); k=d['kernels_ms_per_step']
print('$1', d['value'], d['ms_per_step'], 'single', (d.get('single_job') or {}).get('ms'), 'fold_on', (d.get('twin_fold_on') or {}).get('ms_per_step'), {a:b for a,b in d.items() if a.startswith('verified')}, {a:k[a] for a in list(k)[:6]})"; }
LZ='lz77 or compress_block or many_blocks'
# config 4: the tree (four waves per block), then the emitter with a token per lane
timeout 200 python bench.py --no-cpu-baseline --workload dup8_m1 2>gpurun_out/${T}_err1.txt | tail -1 | tee gpurun_out/${T}_dup8_tree.json | line dup8_tree
timeout 200 python tools/run_variant.py emit bench.py --no-cpu-baseline --workload dup8_m1 2>gpurun_out/${T}_err2.txt | tail -1 | tee gpurun_out/${T}_dup8_emit.json | line dup8_emit
timeout 150 python tools/run_variant.py emit -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -k "$LZ" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_emit.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_emit.log; tail -2 gpurun_out/${T}_tests_emit.log
timeout 60 python tools/lzprof2.py 1 > gpurun_out/${T}_lzprof_tree.txt 2>&1; cat gpurun_out/${T}_lzprof_tree.txt
ZPQ_PROF_DIR=$GRAFT_REPO_ROOT/tools/_variants/emit_prof timeout 60 python tools/lzprof2.py 1 > gpurun_out/${T}_lzprof_emit.txt 2>&1; cat gpurun_out/${T}_lzprof_emit.txt
echo "[$(( $(date +%s) - S0 )) s] config 4"
# headline: the tree, then the literal bytes packed by the token kernel, then the mask evaluator in the segment parse
timeout 300 python bench.py --no-cpu-baseline --workload silesia_x256_m1 2>gpurun_out/${T}_err3.txt | tail -1 | tee gpurun_out/${T}_headline_tree.json | line headline_tree
timeout 300 python tools/run_variant.py pack bench.py --no-cpu-baseline --workload silesia_x256_m1 2>gpurun_out/${T}_err4.txt | tail -1 | tee gpurun_out/${T}_headline_pack.json | line headline_pack
timeout 150 python tools/run_variant.py pack -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -k "$LZ" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_pack.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_pack.log; tail -2 gpurun_out/${T}_tests_pack.log
timeout 300 python tools/run_variant.py maskspec bench.py --no-cpu-baseline --workload silesia_x256_m1 2>gpurun_out/${T}_err5.txt | tail -1 | tee gpurun_out/${T}_headline_maskspec.json | line headline_maskspec
timeout 150 python tools/run_variant.py maskspec -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py -m gpu -k "$LZ" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_maskspec.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_maskspec.log; tail -2 gpurun_out/${T}_tests_maskspec.log
# this round's boundary tests (findBlock, damaged tails) on the chip
timeout 200 python -m pytest tests/test_gpu_segments.py -m gpu -k "findblock or damaged_tail" -x -q -p no:cacheprovider > gpurun_out/${T}_tests_segments.log 2>&1; echo "rc=$?" >> gpurun_out/${T}_tests_segments.log; tail -2 gpurun_out/${T}_tests_segments.log
echo "[$(( $(date +%s) - S0 )) s] done"
for f in 1 2 3 4 5; do [ -s gpurun_out/${T}_err$f.txt ] && { echo "== err$f"; tail -5 gpurun_out/${T}_err$f.txt; }; done
