"""PyTorch fp32 reference of the refine-net / score-net -- TEST INFRASTRUCTURE ONLY (floating-point oracle).

The reference ships the two networks only as opaque TensorRT engines built from un-vendored ONNX files
(tools/cvt_onnx2trt.bash:3-15; call sites detection_6d_foundationpose/src/foundationpose.cpp:206-208,218-220), so the
arithmetic restated here is the PUBLISHED architecture of NVlabs/FoundationPose (learning/models/refine_network.py,
score_network.py, network_modules.py) [EXT] -- PARITY UNPINNED until the real ONNX files are available.
I/O contract from the reference: inputs two NHWC f32 [N,160,160,6] blobs "render_input"/"transf_input", outputs
"trans"[N,3],"rot"[N,3] (foundationpose.cpp:78-81) / "scores"[N,1] (:83, simple_tests/src/test_foundationpose.cpp:24-35).
"""
import math

import numpy as np
import torch
import torch.nn as nn


class ConvBNReLU(nn.Module):
    def __init__(self, cin, cout, k, stride):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride, (k - 1) // 2, bias=True)
        self.bn = nn.BatchNorm2d(cout)

    def forward(self, x):
        return torch.relu(self.bn(self.conv(x)))


class ResnetBasicBlock(nn.Module):
    def __init__(self, c):
        super().__init__()
        self.conv1 = nn.Conv2d(c, c, 3, 1, 1, bias=True)
        self.bn1 = nn.BatchNorm2d(c)
        self.conv2 = nn.Conv2d(c, c, 3, 1, 1, bias=True)
        self.bn2 = nn.BatchNorm2d(c)

    def forward(self, x):
        out = torch.relu(self.bn1(self.conv1(x)))
        out = self.bn2(self.conv2(out))
        return torch.relu(out + x)


def _encode_a():
    return nn.Sequential(ConvBNReLU(6, 64, 7, 2), ConvBNReLU(64, 128, 3, 2), ResnetBasicBlock(128), ResnetBasicBlock(128))


def _encode_ab():
    return nn.Sequential(ResnetBasicBlock(256), ResnetBasicBlock(256), ConvBNReLU(256, 512, 3, 2),
                         ResnetBasicBlock(512), ResnetBasicBlock(512))


class PositionalEmbedding(nn.Module):
    def __init__(self, d_model=512, max_len=400):
        super().__init__()
        pe = torch.zeros(max_len, d_model).float()
        position = torch.arange(0, max_len).float().unsqueeze(1)
        div_term = (torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model)).exp()[None]
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, x):
        return x + self.pe[:, :x.size(1)]


def _trunk(self, A, B):
    bs = len(A)
    x = torch.cat([A, B], dim=0).permute(0, 3, 1, 2)       # NHWC blobs -> NCHW
    x = self.encodeA(x)
    ab = torch.cat((x[:bs], x[bs:]), 1).contiguous()
    ab = self.encodeAB(ab)
    return self.pos_embed(ab.reshape(bs, ab.shape[1], -1).permute(0, 2, 1))


class RefineNet(nn.Module):
    def __init__(self):
        super().__init__()
        self.encodeA, self.encodeAB = _encode_a(), _encode_ab()
        self.pos_embed = PositionalEmbedding()
        self.trans_head = nn.Sequential(nn.TransformerEncoderLayer(512, 4, 512, dropout=0.0, batch_first=True), nn.Linear(512, 3))
        self.rot_head = nn.Sequential(nn.TransformerEncoderLayer(512, 4, 512, dropout=0.0, batch_first=True), nn.Linear(512, 3))

    def forward(self, A, B):
        ab = _trunk(self, A, B)
        return self.trans_head(ab).mean(dim=1), self.rot_head(ab).mean(dim=1)


class ScoreNetMultiPair(nn.Module):
    def __init__(self):
        super().__init__()
        self.encodeA, self.encodeAB = _encode_a(), _encode_ab()
        self.pos_embed = PositionalEmbedding()
        self.att = nn.MultiheadAttention(512, 4, bias=True, batch_first=True)
        self.att_cross = nn.MultiheadAttention(512, 4, bias=True, batch_first=True)
        self.linear = nn.Linear(512, 1)

    def extract_feat(self, A, B):
        ab = _trunk(self, A, B)
        ab, _ = self.att(ab, ab, ab, need_weights=False)
        return ab.mean(dim=1)

    def head(self, feats):
        x = feats[None]
        x, _ = self.att_cross(x, x, x, need_weights=False)
        return self.linear(x).reshape(-1)

    def forward(self, A, B):
        return self.head(self.extract_feat(A, B))


def load_state(model: nn.Module, state: dict) -> nn.Module:
    sd = {k: torch.from_numpy(np.asarray(v)) for k, v in state.items()}
    missing, unexpected = model.load_state_dict(sd, strict=False)
    missing = [m for m in missing if "num_batches_tracked" not in m and not m.endswith("pos_embed.pe")]
    assert not missing and not unexpected, (missing, unexpected)
    return model.eval()


def build(kind: str, state: dict) -> nn.Module:
    return load_state(RefineNet() if kind == "refiner" else ScoreNetMultiPair(), state)
