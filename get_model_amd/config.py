"""Minimal stand-alone reader for the reference's model YAMLs (config/model/GETRegion*.yaml).

Hydra / omegaconf are not required (and not installed on the GPU box): this resolves `${a.b.c}`
interpolations, gives attribute + item access (`cfg.encoder.num_heads`, `**cfg.encoder`) and instantiates
nested `_target_` nodes (the loss components) the way `hydra.utils.instantiate` does.  An omegaconf DictConfig
produced by the reference's own Hydra entry points works too (the models only read attributes / items).
"""
import importlib
import re
from collections.abc import Mapping

import yaml


class AttrDict(dict):
    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError as e:
            raise AttributeError(k) from e

    def __setattr__(self, k, v):
        self[k] = v


_INTERP = re.compile(r"\$\{([^}]+)\}")


def _lookup(root, path):
    cur = root
    for part in path.split("."):
        cur = cur[part]
    return cur


def _resolve(node, root):
    if isinstance(node, Mapping):
        return AttrDict({k: _resolve(v, root) for k, v in node.items()})
    if isinstance(node, list):
        return [_resolve(v, root) for v in node]
    if isinstance(node, str):
        m = _INTERP.fullmatch(node.strip())
        if m:
            return _resolve(_lookup(root, m.group(1)), root)
        return _INTERP.sub(lambda mm: str(_resolve(_lookup(root, mm.group(1)), root)), node)
    return node


def _locate(target):
    mod, _, name = target.rpartition(".")
    return getattr(importlib.import_module(mod), name)


def instantiate(node):
    """Recursively build `_target_` nodes (depth-first), leaving plain containers in place."""
    if isinstance(node, Mapping):
        built = AttrDict({k: instantiate(v) for k, v in node.items() if k != "_target_"})
        if "_target_" in node:
            return _locate(node["_target_"])(**built)
        return built
    if isinstance(node, list):
        return [instantiate(v) for v in node]
    return node


def load_model_yaml(path_or_text, target_prefix=("get_model.model.model.", "get_model_amd.model.model.")):
    """Returns (model_class, cfg) from a model YAML with a top-level `model: {_target_, cfg}` node.
    A reference `_target_` (get_model.model.model.X) is mapped onto this package's class of the same name."""
    text = path_or_text
    if "\n" not in path_or_text:
        with open(path_or_text) as fh:
            text = fh.read()
    raw = yaml.safe_load(text)
    root = _resolve(raw, raw)
    model = root["model"]
    target = model["_target_"]
    if target.startswith(target_prefix[0]):
        target = target_prefix[1] + target[len(target_prefix[0]):]
    cfg = instantiate(model["cfg"])
    return _locate(target), cfg


def build_model(path_or_text):
    cls, cfg = load_model_yaml(path_or_text)
    return cls(cfg)


def region_model_yaml(kind="finetune", num_layers=12, embed_dim=768, num_heads=12, num_motif=283, dropout=0.1):
    """Text of config/model/GETRegion{Finetune,FinetuneATAC,Pretrain}.yaml (schema verbatim, values
    parameterised) -- used by bench.py / tests where the reference tree is absent."""
    name = {"finetune": "GETRegionFinetune", "atac": "GETRegionFinetuneATAC", "pretrain": "GETRegionPretrain"}[kind]
    out_dim = {"finetune": "2", "atac": "1", "pretrain": "${model.cfg.num_motif}"}[kind]
    head = {
        "finetune": "    head_exp:\n      embed_dim: ${model.cfg.embed_dim}\n      output_dim: ${model.cfg.output_dim}\n      use_atac: false\n",
        "atac": "    head_exp:\n      embed_dim: ${model.cfg.embed_dim}\n      output_dim: 1\n      use_atac: false\n",
        "pretrain": "    head_mask:\n      in_features: ${model.cfg.embed_dim}\n      out_features: ${model.cfg.output_dim}\n",
    }[kind]
    lname = {"finetune": "exp", "atac": "atpm", "pretrain": "masked"}[kind]
    if kind == "pretrain":
        loss = f"        {lname}:\n          _target_: torch.nn.MSELoss\n          reduction: \"mean\"\n"
        metrics = '["pearson", "mse", "r2"]'
    else:
        loss = (f"        {lname}:\n          _target_: torch.nn.PoissonNLLLoss\n          reduction: \"mean\"\n"
                f"          log_input: False\n")
        metrics = '["pearson", "spearman", "r2"]'
    return f"""model:
  _target_: get_model.model.model.{name}
  cfg:
    num_regions: 900
    num_motif: {num_motif}
    embed_dim: {embed_dim}
    num_layers: {num_layers}
    num_heads: {num_heads}
    dropout: {dropout}
    output_dim: {out_dim}
    flash_attn: false
    pool_method: "mean"
    region_embed:
      num_features: ${{model.cfg.num_motif}}
      embed_dim: ${{model.cfg.embed_dim}}
    encoder:
      num_heads: ${{model.cfg.num_heads}}
      embed_dim: ${{model.cfg.embed_dim}}
      num_layers: ${{model.cfg.num_layers}}
      drop_path_rate: ${{model.cfg.dropout}}
      drop_rate: 0
      attn_drop_rate: 0
      use_mean_pooling: false
      flash_attn: ${{model.cfg.flash_attn}}
{head}    mask_token:
      embed_dim: ${{model.cfg.embed_dim}}
      std: 0.02
    loss:
      components:
{loss}      weights:
        {lname}: 1.0
    metrics:
      components:
        {lname}: {metrics}
"""
