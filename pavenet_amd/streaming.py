"""Streaming / whole-video inference with a per-frame encoder-memory cache.

The reference's dataset emits one clip per labelled frame as the window
[t - T//2, ..., t, ..., t + T//2] with edge replication at both ends of the video
(opera/datasets/posetrack_video_pose.py:578-623), and ``simple_test`` recomputes backbone +
neck + the 6-layer encoder -- 97 % of the work -- for every frame of every window, although
neighbouring windows share T - 1 frames.  Those stages are per-frame independent (frames live
in the batch dimension, resnet.py:639, OT:21312), so here every frame is encoded ONCE into its
``memory`` slab [S, 256] (22.9 MB at 800x1344) and each window only runs the decoders on T cached
slabs: ~T x fewer backbone / encoder passes.
The decoders' `value_proj` (3 pose-decoder + 2 joint-decoder layers) depends on the frame only as
well: the projected values are cached per frame and layer too ([n_frames, S, 8, 32] each) and the
fused T-frame attention kernels address them through a FRAME TABLE (slab of frame t of window b =
table[b * T + t]) -- no per-window stack copy and no re-projection of the T frames of every window.

Memory per cached frame at 800x1344: 22.9 MB of encoder memory + 5 x 22.9 MB of projected values
(``cache_values=True``, the default) = 137 MB, i.e. 1000 frames = 137 GB of the 288 GB HBM
(23 GB with ``cache_values=False``, which re-projects the T frames of every window instead).

The projected values belong to the slab list ``encode`` returns (``FrameSlabs.values``): ``decode``
uses the frame table only with such a list whose cache covers every slab in it, so slabs kept from
an earlier ``encode``, of another video, or a plain list take the per-window projection instead of
indexing somebody else's cache.  ``encode(frames, into=slabs)`` appends to both.
"""
import torch


class FrameSlabs(list):
    """The per-frame encoder memories of one video (a list of [S, C] tensors) together with what
    was derived from exactly these frames: the decoder layers' projected values
    (``values = (pose-decoder caches [layers], joint-decoder caches [layers])``, each
    [capacity >= len(self), S, 8, 32], rows in list order) and the encoder geometry."""

    def __init__(self):
        super().__init__()
        self.values = None       # None: no cache (padded frames / cache_values=False / host tensors)
        self.n_cached = 0        # frames whose projected values are in `values`
        self.geom = None
        self.levels = None

    def _append_values(self, vals, n_pose, expected_total):
        """Projected values of the frames just appended (one tensor [n, ...] per decoder layer)."""
        n = vals[0].shape[0]
        flat = None if self.values is None else self.values[0] + self.values[1]
        need = self.n_cached + n
        if flat is None or flat[0].shape[0] < need:
            cap = max(need, expected_total, 0 if flat is None else 2 * flat[0].shape[0])
            grown = [v.new_empty((cap,) + tuple(v.shape[1:])) for v in vals]
            if flat is not None:
                for gnew, gold in zip(grown, flat):
                    gnew[:self.n_cached].copy_(gold[:self.n_cached])
            flat = grown
        for c, v in zip(flat, vals):
            c[self.n_cached:need].copy_(v)
        self.values = (flat[:n_pose], flat[n_pose:])
        self.n_cached = need

    def covers(self, indices):
        """The value cache holds every frame of this list, and `indices` address this list."""
        return (self.values is not None and self.n_cached == len(self) and len(self) > 0
                and 0 <= min(indices) and max(indices) < self.n_cached)


class VideoPoseStream:
    """``VideoPoseStream(model, img_meta).infer_video(frames)`` -> one result per frame, equal
    to ``model.simple_test`` on that frame's edge-replicated window."""

    def __init__(self, model, img_meta, encode_chunk=8, decode_chunk=4, cache_values=True):
        self.model = model
        self.head = model.bbox_head
        self.tr = model.bbox_head.transformer
        self.T = self.head.num_frames
        self.meta = img_meta
        self.encode_chunk = encode_chunk
        self.decode_chunk = decode_chunk
        self.cache_values = cache_values
        self._last_levels = None

    @torch.no_grad()
    def encode(self, frames, into=None):
        """frames [n, 3, H, W] on the device -> FrameSlabs of n memory slabs [S, C] (appended to
        `into`, a FrameSlabs of the same video, when given)."""
        from .deform_attn import project_values_hoisted
        slabs = FrameSlabs() if into is None else into
        assert isinstance(slabs, FrameSlabs), 'encode(into=...) takes the FrameSlabs of an earlier encode()'
        nf = frames.shape[0]
        total = len(slabs) + nf
        pose_attn = [l.attentions[-1] for l in self.tr.decoder.layers]
        joint_attn = [l.attentions[-1] for l in self.tr.refine_decoder.layers]
        # a cache must cover ALL slabs of the list: appending to a list that has none stays uncached
        caching = self.cache_values and (len(slabs) == 0 or slabs.values is not None)
        for i in range(0, nf, self.encode_chunk):
            x = frames[i:i + self.encode_chunk]
            n = x.shape[0]
            feats = self.model.extract_feat(x)
            masks, pos, has_padding = self.head.make_masks(feats, [self.meta], frames_per_clip=n)
            memory, _, _, geom = self.tr.encode_frames(feats, masks, pos, has_padding)
            slabs.geom, slabs.levels = geom, [tuple(f.shape[-2:]) for f in feats]
            self._last_levels = slabs.levels
            slabs.extend(memory.unbind(0))
            if caching and not has_padding and memory.is_cuda:
                # value_proj of every decoder layer, once per frame (padded batches mask the value
                # per window and keep the per-window projection)
                # (the joint attention's own projection takes [B, T, S, C]: one "clip" of n frames)
                vals = project_values_hoisted(pose_attn, memory, None) + \
                    project_values_hoisted(joint_attn, memory[None], None)
                slabs._append_values(vals, len(pose_attn), total)
            else:
                caching, slabs.values, slabs.n_cached = False, None, 0
        return slabs

    def _levels_of(self, slab):
        """Level sizes of a plain slab list (no FrameSlabs): those of the last encode() call."""
        assert self._last_levels is not None, 'decode: no encode() has run on this stream yet'
        assert slab.shape[0] == sum(h * w for h, w in self._last_levels), 'decode: slabs of another input size'
        return self._last_levels

    @staticmethod
    def window_indices(n_frames, T):
        """Edge-replicated window of every centre frame (posetrack_video_pose.py:578-623)."""
        h = T // 2
        return [[min(max(c + k, 0), n_frames - 1) for k in range(-h, h + 1)]
                for c in range(n_frames)]

    @torch.no_grad()
    def decode(self, slabs, windows, rescale=False, force_topk_proposals=None,
               force_score_topk=None):
        """Run head + decoders + OKS-NMS on windows of cached slabs (B = len(windows)).
        The two `force_*` index tensors pin the proposal / score top-k selections (parity tests:
        run-to-run rounding noise of the vendor GEMM / conv kernels can swap near-tied members)."""
        T = self.T
        B = len(windows)
        dev = slabs[0].device
        index = [int(i) for w in windows for i in w]
        assert all(len(w) == T for w in windows) and 0 <= min(index) and max(index) < len(slabs), \
            'decode: every window holds T indices into `slabs`'
        memory = torch.stack([slabs[i] for i in index], 0)  # [B*T, S, C]
        metas = [self.meta] * B
        levels = slabs.levels if isinstance(slabs, FrameSlabs) and slabs.levels else self._levels_of(slabs[0])
        geom = slabs.geom if isinstance(slabs, FrameSlabs) and slabs.geom is not None else \
            self.tr.geometry(levels, dev)
        masks, pos, has_padding = self.head.make_masks_from_shapes(B * T, levels, dev, metas)
        mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
        valid_ratios = torch.stack([self.tr.get_valid_ratio(m) for m in masks], 1)
        if valid_ratios.shape[0] != B * T:
            valid_ratios = valid_ratios.expand(B * T, -1, -1)
        encoded = (memory, mask_flatten, valid_ratios, geom)
        kw = {} if force_topk_proposals is None else dict(force_topk_proposals=force_topk_proposals)
        # the frame table indexes the value cache that was built from THESE slabs, or is not used
        if isinstance(slabs, FrameSlabs) and slabs.covers(index) and not has_padding:
            table = torch.tensor(index, dtype=torch.int32, device=dev)
            kw.update(values_projected=slabs.values[0], value_frame_table=table,
                      refine_value_cache=(slabs.values[1], table))
        outs = self.head(None, metas, precomputed=(masks, pos, has_padding, encoded), **kw)
        return self.head.get_bboxes(outs, metas, rescale=rescale, force_score_topk=force_score_topk)

    @torch.no_grad()
    def infer_video(self, frames, rescale=False):
        """frames [N, 3, H, W] -> list of N (bboxes, labels, kpts) tuples (device tensors)."""
        slabs = self.encode(frames)
        wins = self.window_indices(len(slabs), self.T)
        results = []
        for i in range(0, len(wins), self.decode_chunk):
            res = self.decode(slabs, wins[i:i + self.decode_chunk], rescale=rescale)
            results.extend(self.head.results_to_list(res))
        return results
