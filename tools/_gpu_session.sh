cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x -k "stack_and_autoreset or scores or fused or full_size or checkpoint or other_preprocessors or rollouts or debug_reward or capacity or copy_obs or reference_vectors" 2>&1 | tail -4
for k in 1 2; do
timeout 300 python bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('20-step: %.3f M  %.4f ms/step  eps %d' % (d['value']/1e6, d['ms_per_step'], d['config']['episodes_finished']))"
done
timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('400-step: %.3f M  %.4f ms/step  eps %d' % (d['value']/1e6, d['ms_per_step'], d['config']['episodes_finished']), d['roofline']['kernel_alone'])"
timeout 300 python tools/episode_end_probe.py 2>&1 | head -3
