cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for t in ClusterColour; do for L in 32 64; do
timeout 300 python bench.py --no-cpu-baseline --lanes $L --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$t L=$L fused %.3f M  %.3f ms/step; alone %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_alone']['avg_launch_ms']))"
done; done
for t in MoveToCorner; do for L in 8 16 32; do
timeout 300 python bench.py --no-cpu-baseline --lanes $L --steps 240 --task $t-Demo-LoRes4E-v0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); r=d['roofline']; print('$t L=$L fused %.3f M  %.3f ms/step; alone %s' % (d['value']/1e6, d['ms_per_step'], r['kernel_alone']['avg_launch_ms']))"
done; done
