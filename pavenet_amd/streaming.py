"""Streaming / whole-video inference with a per-frame encoder-memory cache.

The reference's dataset emits one clip per labelled frame as the window
[t - T//2, ..., t, ..., t + T//2] with edge replication at both ends of the video
(opera/datasets/posetrack_video_pose.py:578-623), and ``simple_test`` recomputes backbone +
neck + the 6-layer encoder -- 97 % of the work -- for every frame of every window, although
neighbouring windows share T - 1 frames.  Those stages are per-frame independent (frames live
in the batch dimension, resnet.py:639, OT:21312), so here every frame is encoded ONCE into its
``memory`` slab [S, 256] (22.9 MB at 800x1344; a 1000-frame video is 23 GB of the 288 GB HBM)
and each window only runs the decoders on T cached slabs: ~T x fewer backbone / encoder passes.
The decoders' `value_proj` (3 pose-decoder + 2 joint-decoder layers) depends on the frame only as
well: the projected values are cached per frame and layer too ([n_frames, S, 8, 32] each) and the
fused T-frame attention kernels address them through a FRAME TABLE (slab of frame t of window b =
table[b * T + t]) -- no per-window stack copy and no re-projection of the T frames of every window.
"""
import torch


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
        self._vcache = None   # (pose-decoder caches [layers], joint-decoder caches [layers])

    @torch.no_grad()
    def encode(self, frames):
        """frames [n, 3, H, W] on the device -> list of n memory slabs [S, C]."""
        from .deform_attn import project_values_hoisted
        slabs = []
        nf = frames.shape[0]
        pose_attn = [l.attentions[-1] for l in self.tr.decoder.layers]
        joint_attn = [l.attentions[-1] for l in self.tr.refine_decoder.layers]
        cache = None
        for i in range(0, nf, self.encode_chunk):
            x = frames[i:i + self.encode_chunk]
            n = x.shape[0]
            feats = self.model.extract_feat(x)
            masks, pos, has_padding = self.head.make_masks(feats, [self.meta], frames_per_clip=n)
            memory, _, _, geom = self.tr.encode_frames(feats, masks, pos, has_padding)
            self._geom, self._levels = geom, [tuple(f.shape[-2:]) for f in feats]
            slabs.extend(memory.unbind(0))
            if self.cache_values and not has_padding and memory.is_cuda:
                # value_proj of every decoder layer, once per frame (padded batches mask the value
                # per window and keep the per-window projection)
                # (the joint attention's own projection takes [B, T, S, C]: one "clip" of n frames)
                vals = project_values_hoisted(pose_attn, memory, None) + \
                    project_values_hoisted(joint_attn, memory[None], None)
                if cache is None:
                    cache = [v.new_empty((nf,) + tuple(v.shape[1:])) for v in vals]
                for c, v in zip(cache, vals):
                    c[i:i + n].copy_(v)
        self._vcache = None if cache is None else (cache[:len(pose_attn)], cache[len(pose_attn):])
        return slabs

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
        memory = torch.stack([slabs[i] for w in windows for i in w], 0)  # [B*T, S, C]
        metas = [self.meta] * B
        masks, pos, has_padding = self.head.make_masks_from_shapes(B * T, self._levels, dev, metas)
        mask_flatten = torch.cat([m.flatten(1) for m in masks], 1)
        valid_ratios = torch.stack([self.tr.get_valid_ratio(m) for m in masks], 1)
        if valid_ratios.shape[0] != B * T:
            valid_ratios = valid_ratios.expand(B * T, -1, -1)
        encoded = (memory, mask_flatten, valid_ratios, self._geom)
        kw = {} if force_topk_proposals is None else dict(force_topk_proposals=force_topk_proposals)
        if self._vcache is not None and not has_padding:
            table = torch.tensor([i for w in windows for i in w], dtype=torch.int32, device=dev)
            kw.update(values_projected=self._vcache[0], value_frame_table=table,
                      refine_value_cache=(self._vcache[1], table))
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
