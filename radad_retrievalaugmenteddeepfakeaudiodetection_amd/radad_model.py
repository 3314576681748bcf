"""RADADModel / DetectionModel -- inference forward of radad_model.py:9-41 and detection_model.py:9-125 on the GPU.

Module tree and parameter names equal the reference's (`projection_layer.*`, `fuse.*`, `detection_model.model.<i>.*`
with the same nn.Sequential indices), so `load_state_dict` of a reference checkpoint works unchanged.  forward()
runs csrc/proj.hip: the projection (one pass over the neighbour tensor), then `radad_fuse_head_forward` --
fuse Linear over [tpp ; proj] without building the concatenation, and the detection MLP with BatchNorm in its
eval form (running statistics as scale/shift).  Training stays in the reference: a training-mode forward raises.
"""
import ctypes as C

import torch
import torch.nn as nn

from . import _lib
from .projection import ProjectionLayer


class DetectionModel(nn.Module):
    """detection_model.py:9-72: [input_dim] + detection_hidden_dims + [1]; Linear (+ BatchNorm1d | LayerNorm) + ReLU
    + Dropout per hidden layer, Linear alone for the output.  Holds parameters only; RADADModel runs it."""

    def __init__(self, config, input_dim: int):
        super().__init__()
        self.config = config
        self.device = torch.device(getattr(config, "device", "cuda"))
        self.use_batch_norm = bool(getattr(config, "use_batch_norm", True))
        self.use_layer_norm = bool(getattr(config, "use_layer_norm", False))
        self.layers_dims = [int(input_dim)] + [int(d) for d in config.detection_hidden_dims] + [1]
        self.num_layers = len(self.layers_dims) - 1
        if self.num_layers > _lib.HEAD_MAX_LAYERS:
            raise ValueError(f"at most {_lib.HEAD_MAX_LAYERS} Linear layers in the detection head")
        if self.use_layer_norm and not self.use_batch_norm:
            raise NotImplementedError("detection head with LayerNorm: only the BatchNorm (default) and plain variants "
                                      "have a HIP forward")
        layers = []
        for i in range(self.num_layers):                                       # detection_model.py:45-70
            layers.append(nn.Linear(self.layers_dims[i], self.layers_dims[i + 1]))
            if i < self.num_layers - 1:
                if self.use_batch_norm:
                    layers.append(nn.BatchNorm1d(self.layers_dims[i + 1]))
                layers.append(nn.ReLU(inplace=True))
                if float(getattr(config, "detection_dropout", 0.1)) > 0:
                    layers.append(nn.Dropout(float(getattr(config, "detection_dropout", 0.1))))
        self.model = nn.Sequential(*layers)
        for m in self.modules():                                               # :93-105
            if isinstance(m, nn.Linear):
                nn.init.kaiming_uniform_(m.weight, nonlinearity="relu")
                nn.init.zeros_(m.bias)
        self.to(self.device)

    def head_layers(self):
        """[(Linear, BatchNorm1d | None)] in forward order."""
        out, mods = [], list(self.model)
        for i, m in enumerate(mods):
            if isinstance(m, nn.Linear):
                bn = mods[i + 1] if i + 1 < len(mods) and isinstance(mods[i + 1], nn.BatchNorm1d) else None
                out.append((m, bn))
        return out


class RADADModel(nn.Module):
    """radad_model.py:9-41.  forward(neighbor_vecs [B,K,D], tpp_vecs [B,D]) -> logits [B]."""

    def __init__(self, config, tpp_output_dim: int):
        super().__init__()
        self.config = config
        self.device = torch.device(getattr(config, "device", "cuda"))
        self.tpp_output_dim = int(tpp_output_dim)
        self.projection_layer = ProjectionLayer(config, tpp_output_dim)        # :23
        p = int(config.projection_output_dim)
        self.fuse = nn.Linear(self.tpp_output_dim + p, p)                      # :24-26
        self.detection_model = DetectionModel(config, p)                       # :27
        self.to(self.device)
        self._ws = None

    def _check_eval(self, *tensors):
        if self.training and torch.is_grad_enabled():
            raise RuntimeError("RADADModel here is inference-only (HIP forward); call .eval() / torch.no_grad(), "
                               "or train with the reference module and load its state_dict")
        for t in tensors:
            _lib.require_cuda(t, "input")

    def fuse_and_detect(self, tpp_vecs: torch.Tensor, proj: torch.Tensor, return_fused: bool = False):
        """radad_model.py:39-40 given the projection output."""
        self._check_eval(tpp_vecs, proj)
        t = tpp_vecs.detach().contiguous().float()
        pr = proj.detach().contiguous().float()
        B, D = t.shape
        P = pr.shape[1]
        if D != self.tpp_output_dim or pr.shape[0] != B or P != self.fuse.out_features:
            raise ValueError(f"expected tpp [B,{self.tpp_output_dim}] and proj [B,{self.fuse.out_features}], got "
                             f"{tuple(t.shape)} and {tuple(pr.shape)}")
        lib = _lib.load()
        keep = []

        def ptr(x):
            x = x.detach().contiguous().float()
            keep.append(x)
            return x.data_ptr()
        w = _lib.HeadWeights()
        w.wf, w.bf = ptr(self.fuse.weight), ptr(self.fuse.bias)
        layers = self.detection_model.head_layers()
        w.n_layers = len(layers)
        w.dims[0] = P
        for i, (lin, bn) in enumerate(layers):
            w.dims[i + 1] = lin.out_features
            w.lw[i], w.lb[i] = ptr(lin.weight), ptr(lin.bias)
            if bn is not None:      # eval BatchNorm1d: (x - mean) / sqrt(var + eps) * gamma + beta
                scale = bn.weight.detach().float() / torch.sqrt(bn.running_var.float() + bn.eps)
                shift = bn.bias.detach().float() - bn.running_mean.float() * scale
                w.bn_scale[i], w.bn_shift[i] = ptr(scale), ptr(shift)
        n_out = layers[-1][0].out_features if layers else P
        logits = torch.empty((B, n_out), device=t.device, dtype=torch.float32)
        fused = torch.empty((B, P), device=t.device, dtype=torch.float32) if (return_fused or not layers) else None
        need = lib.radad_fuse_head_workspace_bytes(B, D, P)
        if self._ws is None or self._ws.numel() < need or self._ws.device != t.device:
            self._ws = torch.empty(int(need), dtype=torch.uint8, device=t.device)
        with torch.cuda.device(t.device):
            _lib.check(lib.radad_fuse_head_forward(C.byref(w), t.data_ptr(), pr.data_ptr(), B, D, P,
                                                   fused.data_ptr() if fused is not None else None, logits.data_ptr(),
                                                   self._ws.data_ptr(), int(self._ws.numel()), t.device.index,
                                                   _lib.stream_ptr(t.device)), "radad_fuse_head_forward")
        logits = logits.squeeze(-1)                                            # detection_model.py:125
        return (logits, fused) if return_fused else logits

    def forward(self, neighbor_vecs: torch.Tensor, tpp_vecs: torch.Tensor) -> torch.Tensor:
        """radad_model.py:32-41."""
        if neighbor_vecs.device != self.device:
            neighbor_vecs = neighbor_vecs.to(self.device)
        if tpp_vecs.device != self.device:
            tpp_vecs = tpp_vecs.to(self.device)
        self._check_eval(neighbor_vecs, tpp_vecs)
        proj = self.projection_layer(neighbor_vecs)                            # :38
        return self.fuse_and_detect(tpp_vecs, proj)                            # :39-40

    def predict_proba(self, neighbor_vecs: torch.Tensor, tpp_vecs: torch.Tensor) -> torch.Tensor:
        """sigmoid of the logits (detection_model.py:142-155 applied to the fused model)."""
        with torch.no_grad():
            return torch.sigmoid(self.forward(neighbor_vecs, tpp_vecs))
