"""Checkpoint import/export: rename maps of the reference's finetune YAMLs, Lightning prefix,
pretrain -> finetune with strict=False.  CPU only (host code)."""
import torch

from get_model_amd import checkpoint as C
from get_model_amd.config import build_model, region_model_yaml


def test_lightning_round_trip(tmp_path):
    m = build_model(region_model_yaml("finetune", num_layers=2))
    path = C.save_checkpoint(m, str(tmp_path / "last.ckpt"))
    raw = torch.load(path, weights_only=True)
    assert all(k.startswith("model.") for k in raw["state_dict"])
    m2 = build_model(region_model_yaml("finetune", num_layers=2))
    res = C.load_finetune_checkpoint(m2, {"checkpoint": path, "model_key": "state_dict",
                                          "rename_config": C.RENAME_LIGHTNING, "patterns_to_drop": [], "strict": True})
    assert not res.missing_keys and not res.unexpected_keys
    for (k, a), b in zip(m.state_dict().items(), m2.state_dict().values()):
        assert torch.equal(a, b), k


def test_v1_legacy_names_are_renamed(tmp_path):
    # legacy layout (config/finetune/v1_finetune.yaml): blocks.*, fc_norm.*, region_embed.proj.*, head.*, under 'model'
    m = build_model(region_model_yaml("finetune", num_layers=2))
    inverse = [("encoder.blocks.", "blocks."), ("encoder.norm.", "fc_norm."), ("region_embed.embed.", "region_embed.proj."),
               ("head_exp.head.", "head.")]
    legacy = {}
    for k, v in m.state_dict().items():
        for a, b in inverse:
            k = k.replace(a, b)
        legacy[k] = v.clone()
    assert "blocks.0.attn.qkv.weight" in legacy and "head.weight" in legacy and "region_embed.proj.weight" in legacy
    path = str(tmp_path / "v1.pth")
    torch.save({"model": legacy}, path)
    m2 = build_model(region_model_yaml("finetune", num_layers=2))
    res = C.load_finetune_checkpoint(m2, {"checkpoint": path, "model_key": "model", "rename_config": C.RENAME_V1_FINETUNE,
                                          "strict": True})
    assert not res.missing_keys and not res.unexpected_keys
    assert torch.equal(m2.head_exp.head.weight, m.head_exp.head.weight)


def test_pretrain_to_finetune_non_strict_and_drop_patterns(tmp_path):
    pre = build_model(region_model_yaml("pretrain", num_layers=2))
    path = C.save_checkpoint(pre, str(tmp_path / "pre.ckpt"))
    ft = build_model(region_model_yaml("finetune", num_layers=2))
    res = C.load_finetune_checkpoint(ft, {"checkpoint": path, "model_key": "state_dict", "rename_config": {"model.": ""},
                                          "strict": False})
    assert set(res.unexpected_keys) == {"mask_token", "head_mask.weight", "head_mask.bias"}
    assert set(res.missing_keys) == {"head_exp.head.weight", "head_exp.head.bias"}
    assert torch.equal(ft.encoder.blocks[1].mlp.fc2.weight, pre.encoder.blocks[1].mlp.fc2.weight)
    sd = C.rename_lit_state_dict(torch.load(path, weights_only=True)["state_dict"], patterns_to_drop=["head_mask", "mask_token"])
    assert "head_mask.weight" not in sd and "mask_token" not in sd and "cls_token" in sd
    res = C.load_state_dict(ft, torch.load(path, weights_only=True)["state_dict"] | {}, strict=False,
                            patterns_to_drop=["model."])
    assert len(res.missing_keys) == len(ft.state_dict())      # everything dropped -> everything missing
    assert C.extract_state_dict({"model": {"state_dict": {"a": 1}}}) == {"a": 1}
