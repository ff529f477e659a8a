"""Observation preprocessors: the batched mirror of the wrapper stacks in
magical/benchmarks/__init__.py:208-274 (FlattenFrameStack / EagerDictFrameStack + Resize*Observation + ChannelsFirst).

In the reference these are gym.Wrappers around one env; here they are a mixin that changes what
`_observe()` asks the rasteriser for.  A 4-frame stack lives in ONE torch.uint8 tensor
[N, 96, 96, 12] that the raster kernel shifts in place (oldest frame first, newest last):

    LoRes4E / LoRes4A   4 ego / allo frames                                   (:246-256)
    LoResCHW4E          LoRes4E moved to channels-first (a view, no copy)     (:262-268)
    LoRes3EA            [allo_t, ego_t-2, ego_t-1, ego_t]                      (:242-245)
    LoResStack          {'allo': 4 allo frames, 'ego': 4 ego frames}           (:257-261)
"""


def wrap_preproc(env_cls, preproc):
    if preproc is None:
        return env_cls
    if preproc not in ('LoRes4E', 'LoRes4A', 'LoResCHW4E', 'LoRes3EA', 'LoResStack'):
        raise KeyError(preproc)

    class _LoRes(env_cls):
        preproc_name = preproc

        def _build(self):
            import torch
            super()._build()
            mk = lambda: torch.zeros((self.n_envs, 96, 96, 12), dtype=torch.uint8, device=self.device)
            self._stack = mk()
            self._stack_allo = mk() if preproc == 'LoResStack' else None
            self._ones = torch.ones(self.n_envs, dtype=torch.uint8, device=self.device)

        def get_state(self):
            d = super().get_state()
            d['stack'] = self._stack.clone()
            d['stack_allo'] = None if self._stack_allo is None else self._stack_allo.clone()
            return d

        def set_state(self, d):
            super().set_state(d)
            self._stack.copy_(d['stack'])
            if self._stack_allo is not None:
                self._stack_allo.copy_(d['stack_allo'])

        def _observation_space(self):
            from .. import spaces
            box = spaces.Box(0, 255, (12, 96, 96) if preproc == 'LoResCHW4E' else (96, 96, 12), 'uint8')
            return spaces.Dict([('allo', box), ('ego', box)]) if preproc == 'LoResStack' else box

        def _fused_target(self):
            from .. import _native as nat
            if preproc in ('LoRes3EA', 'LoResStack'):
                return None
            return self._stack, (nat.VIEW_ALLO if preproc == 'LoRes4A' else nat.VIEW_EGO), nat.OBS_STACK4

        def _observe(self, fill_all=False, fill_mask=None):
            if self._obs_ready:            # this step's frame is already in the stack (fused step + render)
                self._obs_ready = False
                return self._stack.permute(0, 3, 1, 2) if preproc == 'LoResCHW4E' else self._stack
            if fill_all:
                fill_mask = self._ones
            if preproc == 'LoRes3EA':
                # both halves update the same 12-byte pixels; the launches are ordered by the stream
                self.render_frames(self._stack, view='allo', layout='slot_lo', fill_mask=fill_mask)
                self.render_frames(self._stack, view='ego', layout='stack3_hi', fill_mask=fill_mask)
                return self._stack
            if preproc == 'LoResStack':
                self.render_frames(self._stack_allo, view='allo', layout='stack4', fill_mask=fill_mask)
                self.render_frames(self._stack, view='ego', layout='stack4', fill_mask=fill_mask)
                return {'allo': self._stack_allo, 'ego': self._stack}
            self.render_frames(self._stack, view='allo' if preproc == 'LoRes4A' else 'ego', layout='stack4', fill_mask=fill_mask)
            return self._stack.permute(0, 3, 1, 2) if preproc == 'LoResCHW4E' else self._stack

    _LoRes.__name__ = f'{env_cls.__name__}{preproc}'
    return _LoRes
