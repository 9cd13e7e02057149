"""The caller of the adapter path: the dual-backbone block loop with adapter injection.

Reference: AVE/nets/net_trans_v3.py:669-730 (`MMIL_Net.forward_vit`; AVQA twin net_avst_v2.py:653-705).  For every pair of
(visual Swin block, audio HTS-AT block) the reference runs

    position 1:  res_a, idx = audio_p1[i](f_a, f_v);  res_v, idx = vis_p1[i](f_v, f_a)          (both see the block INPUTS)
                 f_v += drop_path1(norm1(attn(f_v)));  f_v += res_v
                 f_a  = blk_a(f_a);                    f_a += res_a
    position 2:  res_a, idx = audio_p2[i](f_a, f_v);  res_v, idx = vis_p2[i](f_v, f_a)
                 f_v += drop_path2(norm2(mlp(f_v)));   f_v += res_v;   f_a += res_a

calls `idx.squeeze().tolist()` after every site (a host sync each) and appends the lists to `adapter_index_dict`.

Here the two sites of a position run as ONE `AdapterPair` node (two HIP streams, second-use gradients folded), the adapter
residuals are added to the residual streams inside the adapters' output GEMMs where that is safe (`fuse_residual`), the expert
indices stay on the device until `AdapterIndexRecord.to_dict()` is called (one sync per forward instead of 4 per block),
and everything else -- which blocks get adapters, `num_skip`, the 18-vs-6 block alignment of stage 3, downsampling -- follows
the reference.  The backbone blocks themselves are out of scope: any object with the timm Swin-V2
block surface (`_attn, norm1, norm2, mlp, drop_path1, drop_path2`) / the HTS-AT block call (`blk_a(x) -> (x, attn)`) works."""
from __future__ import annotations

from typing import Dict, List, Optional, Sequence

import torch
from torch import nn

from .adapters import AdapterPair, MoEAdapter, _safe_inplace


class AdapterIndexRecord:
    """Expert indices of one forward: record[modality][position] = list (one entry per adapted block) of device tensors.
    `to_dict()` reproduces the reference's `adapter_index_dict` (nested python lists) with one device->host copy."""

    def __init__(self):
        self.entries: Dict[str, Dict[str, List[torch.Tensor]]] = {"audio": {"p1": [], "p2": []}, "video": {"p1": [], "p2": []}}

    def append(self, modality: str, position: str, idx: torch.Tensor):
        self.entries[modality][position].append(idx)

    def to_dict(self) -> Dict[str, Dict[str, list]]:
        flat, where = [], []
        for m, d in self.entries.items():
            for p, lst in d.items():
                for t in lst:
                    flat.append(t.reshape(-1))
                    where.append((m, p, t))
        out = {m: {p: [] for p in d} for m, d in self.entries.items()}
        if not flat:
            return out
        host = torch.cat(flat).cpu()          # the only synchronisation
        off = 0
        for (m, p, t), f in zip(where, flat):
            n = f.numel()
            out[m][p].append(host[off:off + n].reshape(t.shape).squeeze().tolist())      # idx.squeeze().tolist(), :703-704
            off += n
        return out


def align_audio_blocks(vis_blocks: Sequence, aud_blocks: Sequence) -> list:
    """The audio block that runs next to each visual block of a stage, or None (net_trans_v3.py:675-682): equal depth pairs
    them one to one; a deeper visual stage (Swin 18 vs HTS-AT 6) puts audio block j after every `ratio` visual blocks."""
    nv, na = len(vis_blocks), len(aud_blocks)
    if nv == na:
        return list(aud_blocks)
    if na == 0 or nv % na != 0:
        raise ValueError(f"cannot align {na} audio blocks with {nv} visual blocks")
    ratio = nv // na
    out = [None] * nv
    for j, b in enumerate(aud_blocks):
        out[(j + 1) * ratio - 1] = b
    return out


def _to_site(x: torch.Tensor) -> torch.Tensor:
    """(S, N, C) tokens -> the (S, C, N, 1) view the reference hands to the adapters (:694-695)."""
    return x.permute(0, 2, 1).unsqueeze(-1)


def _from_site(r: torch.Tensor) -> torch.Tensor:
    return r.squeeze(-1).permute(0, 2, 1)


class DualBackboneLoop(nn.Module):
    """Runs the block loop of `forward_vit` over two frozen backbones with the MoE adapters injected.

        loop = DualBackboneLoop(model.audio_moe_adapter_blocks_p1, model.vis_moe_adapter_blocks_p1,
                                model.audio_moe_adapter_blocks_p2, model.vis_moe_adapter_blocks_p2, num_skip=opt.num_skip)
        f_v, f_a, record = loop(model.swin.layers, model.htsat.layers, f_v, f_a)
        adapter_index_dict = record.to_dict()

    The adapter lists are shared with the model (same Parameters, same state_dict keys); p1 / p2 may be None when the
    reference's `is_audio_adapter_p1 / p2` flags are off."""

    def __init__(self, audio_p1: Optional[Sequence[nn.Module]], vis_p1: Optional[Sequence[nn.Module]],
                 audio_p2: Optional[Sequence[nn.Module]], vis_p2: Optional[Sequence[nn.Module]], num_skip: int = 1,
                 concurrent: bool = True, fuse_residual: bool = True):
        super().__init__()
        self.fuse_residual = bool(fuse_residual)
        if (audio_p1 is None) != (vis_p1 is None) or (audio_p2 is None) != (vis_p2 is None):
            raise ValueError("a position has adapters for both modalities or for neither")
        self.num_skip, self.concurrent = int(num_skip), bool(concurrent)
        self.p1 = self._pairs(audio_p1, vis_p1)
        self.p2 = self._pairs(audio_p2, vis_p2)

    def _pairs(self, aud, vis):
        if aud is None:
            return None
        if len(aud) != len(vis):
            raise ValueError("audio / visual adapter lists differ in length")
        pairs = []
        for a, v in zip(aud, vis):
            if isinstance(a, MoEAdapter) and isinstance(v, MoEAdapter) and a.variant in ("ave", "avqa") and v.variant in ("ave", "avqa"):
                pairs.append(AdapterPair(a, v, concurrent=self.concurrent))
            else:
                pairs.append(_SequentialPair(a, v))
        return pairs          # a plain list: the sites stay registered in the model only (no duplicate state_dict keys)

    def _adapt(self, pairs, i, f_a, f_v, record, pos, base_a=None, base_v=None):
        """Runs the two sites of one position on (f_a, f_v).  base_*: the residual streams AFTER the backbone half-block; where
        the pair can add into them in place (AdapterPair, and the stream is a fresh sum nobody else reads) the returned
        tensor is the updated stream, otherwise stream + adapter residual is formed here."""
        pair = pairs[i]
        fuse = self.fuse_residual and isinstance(pair, AdapterPair) and base_a is not None
        # in place only into a fresh sum: not one of the adapters' own inputs (saved for their backward), and produced by an
        # addition (whose backward does not read its result)
        # ... that owns its storage and does not overlap the adapters' inputs (adapters._safe_inplace); otherwise base + out
        ok = lambda t: fuse and t.is_contiguous() and t.dtype == f_a.dtype and t.is_cuda and \
            _safe_inplace(t, (f_a, f_v)) and \
            (not t.requires_grad or type(t.grad_fn).__name__.startswith("AddBackward"))
        in_a, in_v = (base_a if ok(base_a) else None), (base_v if ok(base_v) else None)
        if in_a is not None or in_v is not None:
            out_a, idx_a, out_v, idx_v = pair(_to_site(f_a), _to_site(f_v), add_to=(in_a, in_v))
        else:
            out_a, idx_a, out_v, idx_v = pair(_to_site(f_a), _to_site(f_v))
        record.append("audio", pos, idx_a)
        record.append("video", pos, idx_v)
        out_a, out_v = _from_site(out_a), _from_site(out_v)
        if base_a is None:
            return out_a, out_v                           # bare residuals
        return (out_a if in_a is not None else base_a + out_a), (out_v if in_v is not None else base_v + out_v)

    def forward(self, vis_layers, aud_layers, f_v: torch.Tensor, f_a: torch.Tensor):
        record = AdapterIndexRecord()
        i = 0
        for layer_index, (vl, al) in enumerate(zip(vis_layers, aud_layers)):
            for blk, blk_a in zip(vl.blocks, align_audio_blocks(vl.blocks, al.blocks)):
                if blk_a is None:                                                            # :721-723
                    f_v = f_v + blk.drop_path1(blk.norm1(blk._attn(f_v)))
                    f_v = f_v + blk.drop_path2(blk.norm2(blk.mlp(f_v)))
                    continue
                if self.num_skip > 1 and (layer_index + 1) % self.num_skip == 0:             # :687-692  stage without adapters
                    f_v = f_v + blk.drop_path1(blk.norm1(blk._attn(f_v)))
                    f_a, _ = blk_a(f_a)
                    f_v = f_v + blk.drop_path2(blk.norm2(blk.mlp(f_v)))
                    continue
                # The adapters of a position see the streams BEFORE the backbone half-block and their residuals are added AFTER
                # it; the half-blocks do not depend on the adapters, so they run first and the adapter outputs are added into
                # their results inside the output GEMMs (same values as the reference's order of statements).
                if self.p1 is not None:
                    if i >= len(self.p1):
                        raise IndexError("more adapted blocks than position-1 adapter sites")
                    base_v = f_v + blk.drop_path1(blk.norm1(blk._attn(f_v)))
                    base_a, _ = blk_a(f_a)
                    f_a, f_v = self._adapt(self.p1, i, f_a, f_v, record, "p1", base_a, base_v)
                else:
                    f_a, _ = blk_a(f_a)                   # (the reference skips the visual attention half without p1 adapters)
                if self.p2 is not None:
                    if i >= len(self.p2):
                        raise IndexError("more adapted blocks than position-2 adapter sites")
                    base_v = f_v + blk.drop_path2(blk.norm2(blk.mlp(f_v)))
                    f_a, f_v = self._adapt(self.p2, i, f_a, f_v, record, "p2", f_a, base_v)
                else:
                    f_v = f_v + blk.drop_path2(blk.norm2(blk.mlp(f_v)))
                i += 1
            f_v = vl.downsample(f_v)                                                         # :725-727
            if getattr(al, "downsample", None) is not None:
                f_a = al.downsample(f_a)
        return f_v, f_a, record


class _SequentialPair(nn.Module):
    """Two sites called one after the other (the variants `AdapterPair` does not cover, or foreign modules): same return
    tuple as AdapterPair."""

    def __init__(self, site_a, site_b):
        super().__init__()
        self.site_a, self.site_b = site_a, site_b

    def forward(self, x_a, x_b):
        ra = self.site_a(x_a, x_b)
        rb = self.site_b(x_b, x_a)
        return ra[0], ra[1], rb[0], rb[1]
