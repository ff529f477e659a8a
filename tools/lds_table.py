"""GPU probe: LDS footprints (k_step, k_raster) and lanes per env of every task variant (development tool)."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import magical_amd
magical_amd.register_envs()
names = sorted(n for n in magical_amd.ALL_REGISTERED_ENVS if n.endswith('-LoRes4E-v0') and 'DebugReward' not in n)
for name in names:
    e = magical_amd.make(name, n_envs=512, device='cuda:0'); e.reset()
    ls, lr = e._lib.mgx_engine_lds_bytes(e._engine, 0), e._lib.mgx_engine_lds_bytes(e._engine, 1)
    print('%-40s L=%2d  k_step %6d B (%d per CU)  k_raster %6d B (%d per CU)' % (name, e.lanes_per_env, ls, 163840 // ((ls + 1279) // 1280 * 1280), lr, min(5, 163840 // ((lr + 1279) // 1280 * 1280))), flush=True)
    e.close()
