"""Observation preprocessors: the batched mirror of the wrapper stacks in
magical/benchmarks/__init__.py:208-274 (FlattenFrameStack / EagerDictFrameStack + Resize*Observation + ChannelsFirst).

In the reference these are gym.Wrappers around one env; here they are a mixin that changes what
`_observe()` asks the rasteriser for.  A 4-frame stack lives in ONE torch.uint8 tensor
[N, 96, 96, 12] that the raster kernel shifts in place (oldest frame first, newest last):

    LoRes4E / LoRes4A   4 ego / allo frames                                   (:246-256)
    LoResCHW4E          LoRes4E moved to channels-first (a view, no copy)     (:262-268)
    LoRes3EA            [allo_t, ego_t-2, ego_t-1, ego_t]                      (:242-245)
    LoResStack          {'allo': 4 allo frames, 'ego': 4 ego frames}           (:257-261)

LoResCHW4E with `obs_ring=R` (or MGX_OBS_RING=R): the frames are kept as channel planes in a ring
u8[N, R, 3, 96, 96] instead.  A channels-first stack of four consecutive frames is then the contiguous window
ring.view(N, 3R, 96, 96)[:, 3(k-3) : 3(k+1)] of that ring: the rasteriser writes ONE 27.6 KB frame per env and step
(layout MGX_OBS_PLANAR) and nothing is shifted.  When the window reaches the end of the ring its last three frames are
copied to the front (once every R-3 steps), and an env whose episode ended gets its three older slots filled with the
new episode's first frame.  The bytes handed out are those of the in-place stack, and unlike it, an observation stays
valid for at least R-7 further steps of the same episode (until its oldest slot is written again; an auto-reset refills
the three older slots of the envs it resets).
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
            self._ring = None
            if preproc == 'LoResCHW4E' and self.obs_ring:
                self._ring = torch.zeros((self.n_envs, self.obs_ring, 3, 96, 96), dtype=torch.uint8, device=self.device)
                self._ring_k = 2           # slot of the newest frame (none yet)

        # ---- ring of planar frames
        def _ring_next_slot(self):
            """Advance the window by one frame and return the slot u8[N, 3, 96, 96] the new frame goes to."""
            k = self._ring_k + 1
            if k == self.obs_ring:       # wrap: the three frames that stay in the window move to the front
                src = self._ring[:, k - 3:k]
                self._ring[:, 0:3].copy_(src.clone() if k < 6 else src)      # R = 5: source and destination share slot 2
                k = 3
            self._ring_k = k
            return self._ring[:, k]

        def _ring_window(self):
            k = self._ring_k
            return self._ring.view(self.n_envs, 3 * self.obs_ring, 96, 96)[:, 3 * (k - 3):3 * (k + 1)]

        def _ring_fill(self, fill_all, fill_mask):
            """FrameStack.reset(): the envs of the mask see their new first frame four times."""
            import torch
            k = self._ring_k
            old, new = self._ring[:, k - 3:k], self._ring[:, k:k + 1]
            idx = self._fill_idx            # step(): the envs of the mask, known on the host (no ring-sized temporaries then)
            if fill_all or (idx is not None and len(idx) == self.n_envs):
                old.copy_(new.expand_as(old))
            elif idx is not None:
                it = torch.as_tensor(idx, device=self.device)
                self._ring[it, k - 3:k] = self._ring[it, k:k + 1]
            elif fill_mask is not None:
                old.copy_(torch.where(fill_mask.view(-1, 1, 1, 1, 1) != 0, new, old))

        def get_state(self):
            # the snapshot always holds the channels-last stack u8[N, 96, 96, 12] (what self._stack is), whether the frames live
            # there or as planes in a ring: a checkpoint does not depend on obs_ring / MGX_OBS_RING
            d = super().get_state()
            d['stack'] = self._stack.clone() if self._ring is None else self._ring_window().permute(0, 2, 3, 1).contiguous()
            d['stack_allo'] = None if self._stack_allo is None else self._stack_allo.clone()
            return d

        def set_state(self, d):
            super().set_state(d)
            want = (self.n_envs, 96, 96, 12)
            if tuple(d['stack'].shape) != want:
                raise ValueError(f"snapshot frame stack has shape {tuple(d['stack'].shape)}, expected {want} (channels last)")
            if self._ring is not None:
                self._ring_k = 3
                self._ring[:, 0:4].copy_(d['stack'].permute(0, 3, 1, 2).reshape(self.n_envs, 4, 3, 96, 96))
                return
            self._stack.copy_(d['stack'])
            if self._stack_allo is not None:
                self._stack_allo.copy_(d['stack_allo'])

        def _observation_space(self):
            from .. import spaces
            box = spaces.Box(0, 255, (12, 96, 96) if preproc == 'LoResCHW4E' else (96, 96, 12), 'uint8')
            return spaces.Dict([('allo', box), ('ego', box)]) if preproc == 'LoResStack' else box

        # ---- terminal observations (BaseEnv(terminal_observation=True)): the stacks take the post-step frame of EVERY env -- for the
        # finished envs that is the observation their episode ends with -- and the envs that did not finish get their stacks back after
        # the reset's fill pass has shifted them a second time
        def _stacks(self):
            return [self._stack] if self._stack_allo is None else [self._stack_allo, self._stack]

        def _terminal_begin(self, idx):
            import torch
            it = torch.as_tensor(idx, device=self.device)
            obs = self._observe()
            term = {k: v[it].clone() for k, v in obs.items()} if isinstance(obs, dict) else obs[it].clone()
            saved = None if len(idx) == self.n_envs else [t.clone() for t in self._stacks()]
            return term, saved

        def _terminal_end(self, saved, done_dev):
            import torch
            if saved is not None:
                keep = done_dev.view(-1, 1, 1, 1) != 0
                for t, sv in zip(self._stacks(), saved):
                    t.copy_(torch.where(keep, t, sv))

        def _fused_target(self):
            from .. import _native as nat
            if preproc == 'LoRes3EA':         # the launches of _observe(), in its order
                return [(self._stack, nat.VIEW_ALLO, nat.OBS_SLOT_LO), (self._stack, nat.VIEW_EGO, nat.OBS_STACK3_HI)]
            if preproc == 'LoResStack':
                return [(self._stack_allo, nat.VIEW_ALLO, nat.OBS_STACK4), (self._stack, nat.VIEW_EGO, nat.OBS_STACK4)]
            if self._ring is not None:
                return self._ring_next_slot(), nat.VIEW_EGO, nat.OBS_PLANAR
            return self._stack, (nat.VIEW_ALLO if preproc == 'LoRes4A' else nat.VIEW_EGO), nat.OBS_STACK4

        def _observe(self, fill_all=False, fill_mask=None):
            if self._obs_ready:            # this step's frame is already in the stack (fused step + render)
                self._obs_ready = False
                if self._ring is not None:
                    return self._ring_window()
                if preproc == 'LoResStack':
                    return {'allo': self._stack_allo, 'ego': self._stack}
                return self._stack.permute(0, 3, 1, 2) if preproc == 'LoResCHW4E' else self._stack
            if self._ring is not None:
                self.render_frames(self._ring_next_slot(), view='ego', layout='planar')
                self._ring_fill(fill_all, fill_mask)
                return self._ring_window()
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
