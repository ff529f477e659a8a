"""Observation preprocessors: the batched mirror of the wrapper stacks in
magical/benchmarks/__init__.py:208-274 (FlattenFrameStack + ResizeObservation + ChannelsFirst).

In the reference these are gym.Wrappers around one env; here they are a mixin that changes what
`_observe()` asks the rasteriser for.  The 4-frame stack lives in ONE torch.uint8 tensor
[N, 96, 96, 12] that the raster kernel shifts in place (oldest frame first, newest last).
"""


def wrap_preproc(env_cls, preproc):
    if preproc is None:
        return env_cls
    view = {'LoRes4E': 'ego', 'LoRes4A': 'allo', 'LoResCHW4E': 'ego'}[preproc]
    chw = preproc == 'LoResCHW4E'

    class _LoRes4(env_cls):
        obs_view = view
        channels_first = chw

        def _build(self):
            import torch
            super()._build()
            self._stack = torch.zeros((self.n_envs, 96, 96, 12), dtype=torch.uint8, device=self.device)

        def _observe(self, fill_all=False, fill_mask=None):
            import torch
            if fill_all:
                fill_mask = torch.ones(self.n_envs, dtype=torch.uint8, device=self.device)
            self.render_frames(self._stack, view=self.obs_view, layout='stack4', fill_mask=fill_mask)
            return self._stack.permute(0, 3, 1, 2) if self.channels_first else self._stack

    _LoRes4.__name__ = f'{env_cls.__name__}{preproc}'
    return _LoRes4
