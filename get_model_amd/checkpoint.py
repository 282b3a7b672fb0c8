"""Checkpoint import / export for the region models.

Mirror of the reference helpers `get_model/utils.py:147-227` and of the sequence `RegionLitModel.get_model` runs
(`run_region.py:311-357`, non-LoRA branches), same names and argument meaning, so the finetune YAMLs
(`config/finetune/{lightning,v1_finetune,v1_pretrain}.yaml`: checkpoint, model_key, rename_config, patterns_to_drop,
strict) apply unchanged.  Two deliberate differences: files are read with `torch.load(weights_only=True)` (nothing
in a checkpoint is executed), and `https://` paths are refused -- there is no network on the training hosts.
"""
import torch

# the reference's shipped rename maps (config/finetune/*.yaml), for callers without the YAML at hand
RENAME_LIGHTNING = {"model.": ""}
RENAME_V1_FINETUNE = {"blocks.": "encoder.blocks.", "fc_norm.": "encoder.norm.", "encoder.head.": "head_mask.",
                      "encoder.region_embed": "region_embed", "region_embed.proj.": "region_embed.embed.",
                      "encoder.cls_token": "cls_token", "head.": "head_exp.head."}
RENAME_V1_PRETRAIN = {"encoder.head.": "head_mask.", "encoder.region_embed": "region_embed",
                      "region_embed.proj.": "region_embed.embed.", "encoder.cls_token": "cls_token"}


def load_checkpoint(checkpoint_path, model_key=None):
    """Read a checkpoint file; pick the first of `model_key`'s '|'-separated keys that exists (utils.py:147-168)."""
    if str(checkpoint_path).startswith("http"):
        raise RuntimeError("load_checkpoint: remote checkpoints are not fetched here; download the file first")
    checkpoint = torch.load(checkpoint_path, map_location="cpu", weights_only=True)
    if model_key is not None:
        for key in model_key.split("|"):
            if key in checkpoint:
                return checkpoint[key]
    return checkpoint


def extract_state_dict(checkpoint_model):
    """Unwrap {'model': ...} and then {'state_dict': ...} if present (utils.py:182-187)."""
    if "model" in checkpoint_model:
        checkpoint_model = checkpoint_model["model"]
    if "state_dict" in checkpoint_model:
        checkpoint_model = checkpoint_model["state_dict"]
    return checkpoint_model


def rename_state_dict(state_dict, rename_config):
    """Apply every pattern -> replacement of rename_config, in its order, to every key (utils.py:190-201)."""
    if rename_config is None:
        return state_dict
    out = {}
    for key, value in state_dict.items():
        for pattern, replacement in rename_config.items():
            key = key.replace(pattern, replacement)
        out[key] = value
    return out


def rename_lit_state_dict(state_dict, patterns_to_drop=()):
    """Strip Lightning's 'model.' and drop keys containing any of patterns_to_drop (utils.py:171-179)."""
    out = {}
    for key, value in state_dict.items():
        new_key = key.replace("model.", "")
        if not any(p in new_key for p in patterns_to_drop):
            out[new_key] = value
    return out


def load_state_dict(model, state_dict, strict=True, patterns_to_drop=()):
    """model.load_state_dict after removing keys that contain a dropped pattern (utils.py:223-227)."""
    for pattern in patterns_to_drop:
        state_dict = {k: v for k, v in state_dict.items() if pattern not in k}
    return model.load_state_dict(state_dict, strict=strict)


def load_finetune_checkpoint(model, finetune_cfg):
    """run_region.py:311-357 without LoRA: checkpoint -> model_key -> unwrap -> rename -> (drop) -> load."""
    get = (lambda k, d=None: finetune_cfg.get(k, d)) if isinstance(finetune_cfg, dict) else \
        (lambda k, d=None: getattr(finetune_cfg, k, d))
    if get("checkpoint") is None:
        return None
    sd = load_checkpoint(get("checkpoint"), model_key=get("model_key"))
    sd = rename_state_dict(extract_state_dict(sd), get("rename_config"))
    return load_state_dict(model, sd, strict=bool(get("strict", True)), patterns_to_drop=get("patterns_to_drop") or ())


def save_checkpoint(model, path, lightning=True, extra=None):
    """Export in the layout the reference reads back: {'state_dict': {'model.<key>': tensor}} (what Lightning writes
    for LitModel.model, config/finetune/lightning.yaml) or a bare state_dict."""
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    obj = {"state_dict": {"model." + k: v for k, v in sd.items()}} if lightning else sd
    if extra:
        obj.update(extra)
    torch.save(obj, path)
    return path
