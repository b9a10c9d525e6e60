"""CPU oracle of the temporal mode -- TEST INFRASTRUCTURE (see oracle/__init__.py).

PARITY UNPINNED against the reference: its carried-state branch cannot run (`is_first` is always True, and
the `else` branch of MOTRTrack.forward head.py:206-221 feeds N+nq reference boxes with nq content embeddings into the
decoder, head.py:1055-1064 vs :1104-1113, which raises).  This file states the spec the build implements,
assembled from the pieces that ARE pinned against the reference (decoder, ID loop, isolated QIM update) in the order the
dead branch and upstream MOTR prescribe:

  frame t, one sequence, memory of n live tracks (content embedding, position embedding, reference box, id, misses):
  1. detect queries as shipped: top-k tokens -> content = features, reference = enc boxes, position = pos2posemb
     (head.py:1048-1054,1104-1108);
  2. decoder rows = [tracks | detect queries] (order of head.py:1059,1064): content embedding of a track = its decoder
     output of the previous frame (`track_embed`, head.py:1110-1111; upstream motr.py:545-577 `output_embedding`),
     position = QIM-updated `query_pos`, reference = `ref_pts`;
  3. scores = sigmoid(logits).max; ID loop of RuntimeTrackerBase.update (head.py:1232-1243) on carried ids / counters;
     the id counter only moves at births (upstream motr.py:303-325);
  4. rows with id >= 0 (query order) -> `_update_track_embedding` (qim.py:251-301) with ref_pts / query_pos = what the
     row was decoded with, output_embedding = decoder output, pred_boxes = refined boxes -> the next frame's memory,
     truncated to n_max slots.
"""
from __future__ import annotations

import torch

from . import track_oracle as O


class TemporalOracle:
    def __init__(self, sd, arch, n_max, conf=0.25, content="decoder_output"):
        """content: what a carried track's CONTENT embedding is in the next frame -- "decoder_output" (upstream MOTR, the default) or
        "class_embed": the fork's own visible design, denoising_class_embed.weight[argmax of the track's class scores]
        (head.py:888-900, passed on as track_embed :917-919 and concatenated in front of the detect queries :1109-1110)."""
        assert content in ("decoder_output", "class_embed")
        self.content = content
        self.sd, self.arch, self.n_max, self.conf = sd, arch, n_max, conf
        self.d = f"model.{len(arch.layers)}.decoder"
        self.t = f"model.{len(arch.layers)}.track_embed"
        self.reset()

    def reset(self):
        z = lambda *s: torch.zeros(*s)
        self.embed, self.qpos, self.ref = z(0, 256), z(0, 256), z(0, 4)
        self.ids, self.dis, self.max_obj_id = [], [], 0

    @torch.no_grad()
    def step(self, x, orig_hw=None):
        """x [1, 3, H, W] in [0, 1].  Returns the frame's rows / ids and leaves the memory updated."""
        sd, arch = self.sd, self.arch
        feats, shapes = O.encoder_input(O.backbone_neck(x, sd, arch), sd, self.d)
        di = O.decoder_input(feats, shapes, sd, self.d, arch.nq)
        n = self.embed.shape[0]
        embed = torch.cat([self.embed, di["embed"][0]])[None]
        ref_logit = torch.cat([self.ref, di["refer_bbox_logit"][0]])[None]
        qpos = torch.cat([self.qpos, di["query_pos"][0]])[None]
        boxes, logits, hs = O.decoder(embed, ref_logit, feats, shapes, qpos, sd, self.d, arch)
        scores = logits[0].sigmoid().max(-1).values
        ids0 = self.ids + [-1] * arch.nq
        dis0 = self.dis + [0] * arch.nq
        ids, dis, self.max_obj_id = O.assign_ids_loop(scores, ids0, dis0, self.max_obj_id)
        active = [i for i, v in enumerate(ids) if v >= 0]
        out = dict(scores=scores, boxes=boxes[0], logits=logits[0], ids=torch.tensor(ids), dis=torch.tensor(dis), n_in=n,
                   active=active, hs=hs[0])
        y = torch.cat((boxes[0], logits[0].sigmoid()), -1)
        rows, tid = O.postprocess(y, logits[0], torch.tensor(ids), self.conf, orig_hw=orig_hw)
        out.update(rows=rows, track_id=tid, n_overflow=max(len(active) - self.n_max, 0))
        keep = active[:self.n_max]
        if keep:
            k = torch.tensor(keep)
            qf, new_ref = O.qim_update_track_embedding(ref_logit[0, k], hs[0, k], qpos[0, k], boxes[0, k], sd, self.t, arch.nh)
            self.embed, self.qpos, self.ref = hs[0, k].clone(), qf, new_ref
            if self.content == "class_embed":
                self.embed = sd[self.d + ".denoising_class_embed.weight"][logits[0, k].argmax(-1)].clone()
        else:
            self.embed, self.qpos, self.ref = torch.zeros(0, 256), torch.zeros(0, 256), torch.zeros(0, 4)
        self.ids, self.dis = [ids[i] for i in keep], [dis[i] for i in keep]
        return out
