"""DGCNN context embedder in TRAINING mode (models/pytorch_gcn.py:81-107 per-point, :143-188 global), differentiable
through HIP kernels: k-NN graph (inference kernel; indices carry no gradient), the edge convolution split by linearity into two
per-point products of the training Linear, BatchNorm with batch statistics + LeakyReLU + max over neighbours (csrc/train_edge.hip),
conv5 + BatchNorm1d, the output MLP.  BatchNorm running statistics are updated as torch does in train mode.

torch on activations: data movement only (dense copies for the k-NN kernel, cat of the four level outputs).
"""
import torch

from . import engine
from . import train_ops as T


def dgcnn_embed(emb, pts):
    """pts [B, M, C_in] -> [B, M, E] (DGCNNembedder) or [B, E] (DGCNNembedderGlobal), with autograd to every embedder parameter."""
    B, M, Cin = pts.shape
    rows = B * M
    k = emb.n_neighbors
    f = T.to_panel(pts.reshape(rows, Cin).to(torch.float32))
    width = Cin
    offs = (torch.arange(B, device=pts.device, dtype=torch.int32) * M)[:, None, None]
    outs = []
    for conv, bn in ((emb.conv1, emb.bn1), (emb.conv2, emb.bn2), (emb.conv3, emb.bn3), (emb.conv4, emb.bn4)):
        W = conv[0].weight
        Co = W.shape[0]
        W = W.reshape(Co, 2 * width)
        with torch.no_grad():
            idx = engine.op_knn(f[:rows, :width].reshape(B, M, width).contiguous(), k)          # [B, M, k] indices inside the scene
            idx = (idx + offs).reshape(rows, k).contiguous()                                        # global rows
        wa, wb = W[:, :width], W[:, width:]
        pq = T.linear_act([f], [width], torch.cat((wa, wb - wa), 0), None, rows)                   # [P | Q], models/pytorch_gcn.py:40-44 by linearity
        f = T.edge_bn_max(pq, bn, idx, rows, Co, k)
        width = Co
        outs.append(f)
    cat = torch.cat(outs, -1)                                                                       # 64 + 64 + 128 + 256 columns, no padding inside
    y5 = T.linear_act([cat], [cat.shape[1]], emb.conv5[0].weight.reshape(emb.conv5[0].weight.shape[0], -1), None, rows)
    t = T.edge_bn_max(y5, emb.bn5, None, rows, y5.shape[1], 1)
    if emb.is_global:
        pooled = T.pool_max_mean(t, B, M, t.shape[1])                                              # [B, 1024] = [max | mean] over the scene
        return T.mlp_forward(emb.out_mlp, pooled, "GELU")
    y = T.mlp_panels(emb.out_mlp, [t], [t.shape[1]], rows, "GELU")
    return T.from_panel(y, rows, emb.out_mlp.out_layer.out_features).reshape(B, M, -1)
