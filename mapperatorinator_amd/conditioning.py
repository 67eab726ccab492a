"""Conditioning embedders of the Mapperatorinator wrapper (difficulty / mapper style / song position), host side.

The reference turns each conditioning input into one vector per batch row, repeats it over the encoder frames,
concatenates it to the mel frames and sends the result through `encoder_embedder` (modeling_mapperatorinator.py:395-414) --
or, with project_encoder_input = false ('Tiger14n/ropewhisper-*', V30 / V31), straight into the backbone's conv1 as extra
input channels (`channels()`; mh_cond_channels writes them beside the mel channels).
A vector that is constant along the frames contributes a constant to every frame of its row:

    encoder_embedder([mel | cond]) = mel @ W[:, :n_mels].T + (cond @ W[:, n_mels:].T + b)

so the device sees ONE extra fp32 row vector per chunk (`row_bias`, B x d_model) and the mel GEMM keeps its shape; the
embedders themselves are a few hundred flops per row and stay on the host (eval mode: the reference's Dropout layers
are the identity).  Modules restated: DifficultyEmbedder (:462-515: Gaussian RBF over difficulty / 10 -> Linear ->
LayerNorm -> ReLU -> Linear -> LayerNorm), MapperStyleEmbedder (:518-576: table row, -1 and out-of-range ids -> the
default row, LayerNorm), SongPositionEmbedder (:579-660: RBF of start and end position -> Linear -> LayerNorm -> ReLU
-> Linear -> LayerNorm), LabelEmbedder (:446-459: table row).
"""
from __future__ import annotations

from typing import Optional

import torch
import torch.nn.functional as F


def _rbf(x: torch.Tensor, centers: torch.Tensor, widths: torch.Tensor) -> torch.Tensor:
    return torch.exp(-(x.unsqueeze(-1) - centers.view(1, -1)).pow(2) / (2 * widths.view(1, -1).pow(2)))


def _mlp(x, sd, prefix):
    """Linear -> LayerNorm -> ReLU -> (Dropout) -> Linear -> LayerNorm: Sequential indices 0, 1, 4, 5."""
    h = F.linear(x, sd[prefix + ".0.weight"], sd[prefix + ".0.bias"])
    h = F.layer_norm(h, h.shape[-1:], sd[prefix + ".1.weight"], sd[prefix + ".1.bias"])
    h = F.linear(torch.relu(h), sd[prefix + ".4.weight"], sd[prefix + ".4.bias"])
    return F.layer_norm(h, h.shape[-1:], sd[prefix + ".5.weight"], sd[prefix + ".5.bias"])


class ConditioningEmbedders:
    """Built from the wrapper's state_dict; which embedders exist is read off the keys present."""

    def __init__(self, state_dict: dict, n_mels: int, max_difficulty: float = 10.0):
        sd = {k: v.detach().to(torch.float32).cpu() for k, v in state_dict.items()
              if k.split(".")[0] in ("style_embedder", "difficulty_embedder", "mapper_embedder", "song_pos_embedder")}
        self.sd = sd
        self.n_mels = int(n_mels)
        self.max_difficulty = float(max_difficulty)
        self.has_style = "style_embedder.embedding_table.weight" in sd
        self.has_difficulty = "difficulty_embedder.basis_centers" in sd
        self.has_mapper = "mapper_embedder.embedding.weight" in sd
        self.has_song_position = "song_pos_embedder.basis_centers" in sd
        self.as_channels = "encoder_embedder.weight" not in state_dict
        if self.as_channels:
            # project_encoder_input = false (configs/model/whisper_small_v2.yaml, varwhisper_*_v3.yaml): the vectors are
            # concatenated to the mel frames as INPUT CHANNELS of the backbone's conv1 (modeling_mapperatorinator.py:201-202,211;
            # configuration_mapperatorinator.py:104 num_mel_bins = n_mels + cond_size) -- `channels()` instead of `row_bias()`
            c1 = state_dict.get("transformer.model.encoder.conv1.weight")
            self.cond_size = int(c1.shape[1]) - self.n_mels if c1 is not None else 0
            if (self.cond_size > 0) != bool(sd):
                raise ValueError(f"conv1 takes {self.cond_size} conditioning channels but the state dict carries "
                                 f"{'no' if not sd else 'some'} conditioning embedders")
            self.w_cond, self.bias = torch.zeros(0, 0), torch.zeros(0)
            return
        w = state_dict["encoder_embedder.weight"].detach().to(torch.float32).cpu()
        self.w_cond = w[:, self.n_mels:].contiguous()          # (d_model, cond_size)
        self.bias = state_dict["encoder_embedder.bias"].detach().to(torch.float32).cpu()
        self.cond_size = self.w_cond.shape[1]

    @property
    def active(self) -> bool:
        return self.cond_size > 0

    def vectors(self, batch: int, beatmap_idx=None, difficulty=None, mapper_idx=None, song_position=None) -> Optional[torch.Tensor]:
        """(B, cond_size) fp32 in the reference's concatenation order: style | difficulty | mapper | song position."""
        sd, parts = self.sd, []
        if self.has_style:
            table = sd["style_embedder.embedding_table.weight"]
            idx = torch.full((batch,), table.shape[0] - 1, dtype=torch.long) if beatmap_idx is None else torch.as_tensor(beatmap_idx).long().cpu()
            parts.append(table[idx])                           # `num_classes` = the "no style" row (:389-392)
        if self.has_difficulty:
            if difficulty is None:
                raise ValueError("this model has a difficulty embedder: model_kwargs['difficulty'] is required")
            x = torch.as_tensor(difficulty, dtype=torch.float32).cpu().reshape(batch) / self.max_difficulty
            parts.append(_mlp(_rbf(x, sd["difficulty_embedder.basis_centers"], sd["difficulty_embedder.basis_widths"]),
                              sd, "difficulty_embedder.difficulty_proj"))
        if self.has_mapper:
            if mapper_idx is None:
                raise ValueError("this model has a mapper embedder: model_kwargs['mapper_idx'] is required")
            table = sd["mapper_embedder.embedding.weight"]
            n = table.shape[0] - 1                             # rows 0..n-1 = mappers, row n = default style
            idx = torch.as_tensor(mapper_idx).long().cpu().reshape(batch)
            idx = torch.where(idx == -1, torch.full_like(idx, n), idx).clamp(0, n)
            e = table[idx]
            parts.append(F.layer_norm(e, e.shape[-1:], sd["mapper_embedder.layer_norm.weight"], sd["mapper_embedder.layer_norm.bias"]))
        if self.has_song_position:
            if song_position is None:
                raise ValueError("this model has a song-position embedder: model_kwargs['song_position'] is required")
            pos = torch.as_tensor(song_position, dtype=torch.float32).cpu().reshape(batch, 2)
            c, wd = sd["song_pos_embedder.basis_centers"], sd["song_pos_embedder.basis_widths"]
            parts.append(_mlp(torch.cat([_rbf(pos[:, 0], c, wd), _rbf(pos[:, 1], c, wd)], 1), sd, "song_pos_embedder.position_proj"))
        if not parts:
            return None
        out = torch.cat(parts, -1)
        if out.shape[1] != self.cond_size:
            raise ValueError(f"conditioning vectors have {out.shape[1]} columns, encoder_embedder expects {self.cond_size}")
        return out

    def channels(self, cond: torch.Tensor, storage_dtype: torch.dtype) -> torch.Tensor:
        """(B, cond_size) fp32, rounded to the storage dtype (the reference's embedder modules run in the model's dtype and the
        concatenated frames are conv1's operand): what mh_cond_channels writes beside the mel channels."""
        if not self.as_channels:
            raise ValueError("this model projects its encoder input: the conditioning enters as row_bias()")
        return cond.to(storage_dtype).to(torch.float32)

    def row_bias(self, cond: torch.Tensor, storage_dtype: torch.dtype) -> torch.Tensor:
        """(B, d_model) fp32: cond @ W_cond.T + bias with cond and W_cond rounded to the storage dtype first (they are
        GEMM operands in the reference's concatenated form), accumulated in fp32."""
        if self.as_channels:
            raise ValueError("this model has no encoder_embedder: the conditioning enters as conv1 channels()")
        c = cond.to(storage_dtype).to(torch.float32)
        w = self.w_cond.to(storage_dtype).to(torch.float32)
        return c @ w.t() + self.bias
