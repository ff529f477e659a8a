"""Is the rasteriser's launch time bimodal with the address of the observation tensor?  One process per trial: data_ptr of the stack and
of the pose blob, k_raster alone (HIP events around 40 launches) (development tool)."""
import sys, os, subprocess, json
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
if len(sys.argv) > 1 and sys.argv[1] == 'one':
    sys.path.insert(0, ROOT)
    import numpy as np, torch
    import magical_amd
    pad = int(sys.argv[2])
    junk = torch.empty(pad, dtype=torch.uint8, device='cuda:0') if pad else None     # shifts what the caching allocator hands out next
    env = magical_amd.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=4096, device='cuda:0')
    env.reset()
    tape = torch.as_tensor(np.random.RandomState(0).randint(0, 18, size=(60, 4096)).astype(np.int32), device='cuda:0')
    for s in range(60):
        env.step(tape[s])
    stack = env._stack
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(40):
        env.render_frames(stack, view='ego', layout='stack4')
    b.record(); torch.cuda.synchronize()
    print(json.dumps({'pad': pad, 'stack_ptr': hex(stack.data_ptr()), 'mod_2M': stack.data_ptr() % (2 << 20), 'mod_1G': stack.data_ptr() % (1 << 30), 'raster_ms': a.elapsed_time(b) / 40}))
else:
    for pad in (0, 0, 0, 1 << 20, 3 << 20, 64 << 20, 257 << 20, 0, 1 << 20):
        out = subprocess.run([sys.executable, __file__, 'one', str(pad)], capture_output=True, text=True).stdout.strip().splitlines()
        print(out[-1] if out else 'no output')
