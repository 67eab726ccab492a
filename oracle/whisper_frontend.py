"""TEST INFRASTRUCTURE ONLY -- CPU restatement of the Whisper-style conv front-end.

Follows HF `WhisperEncoder.forward` (transformers models/whisper/modeling_whisper.py: conv1 k3 p1 -> gelu ->
conv2 k3 s2 p1 -> gelu -> permute -> + embed_positions) which the reference's forks repeat verbatim
(osuT5/osuT5/model/custom_transformers/modeling_varwhisper.py:779-780,813-816; no position table there).
Pinned: tests/golden/whisper_frontend.npz is written by oracle/make_golden.py:whisper_frontend_case from the imported
reference `VarWhisperEncoder` (`out_var`, no position table) and the installed HF `WhisperEncoder` (`out`, `pos`);
tests/test_oracle_pinned.py checks this restatement against both.
"""
import torch
import torch.nn.functional as F


def whisper_frontend(x_bcl, w1, b1, w2, b2, pos=None, rounding=None):
    """x (B, C, L) fp32.  rounding='bf16' restates the bf16 storage contract (operands rounded, fp32 accumulate)."""
    r = (lambda t: t.to(torch.bfloat16).float()) if rounding == "bf16" else (lambda t: t)
    y = F.gelu(F.conv1d(r(x_bcl), r(w1), r(b1), padding=1))
    z = F.gelu(F.conv1d(r(y), r(w2), r(b2), stride=2, padding=1))
    z = z.permute(0, 2, 1)
    if pos is not None:
        z = z + r(pos)
    return z
