"""magical_amd -- MI355X-native batched step() engine for the MAGICAL benchmark.

Drop-in for the hot path of qxcv/magical (BaseEnv.step: 10 rigid-body substeps + egocentric
render + LoRes4E preprocessing), keeping its name surface (magical/__init__.py):

    import magical_amd as magical
    magical.register_envs()
    env = magical.make('MoveToCorner-Demo-LoRes4E-v0', n_envs=4096, device='cuda:0')
    obs = env.reset()
    obs, rew, done, info = env.step(actions)      # info['eval_score']

The compute path is hand-written HIP (magical_amd/csrc) behind the C ABI in include/mgx.h;
there is no CPU fallback.
"""
from magical_amd.benchmarks import (ALL_REGISTERED_ENVS, AVAILABLE_PREPROCESSORS, DEMO_ENVS_TO_TEST_ENVS_MAP,  # noqa: F401
                                    EnvName, make, register_envs, update_magical_env_name)
from magical_amd.reference_demos import try_download_demos  # noqa: F401
from magical_amd.saved_trajectories import (load_demos, preprocess_demos_with_wrapper,  # noqa: F401
                                            splice_in_preproc_name)
from magical_amd.version import __version__  # noqa: F401

# (magical/__init__.py:2-8 exports exactly: ALL_REGISTERED_ENVS, AVAILABLE_PREPROCESSORS, DEMO_ENVS_TO_TEST_ENVS_MAP, register_envs,
# try_download_demos, load_demos, preprocess_demos_with_wrapper, splice_in_preproc_name, __version__; the rest is this package's own)
__all__ = ['DEMO_ENVS_TO_TEST_ENVS_MAP', 'register_envs', 'make', 'EnvName', 'ALL_REGISTERED_ENVS',
           'AVAILABLE_PREPROCESSORS', 'update_magical_env_name', 'try_download_demos', 'load_demos',
           'preprocess_demos_with_wrapper', 'splice_in_preproc_name', '__version__']
