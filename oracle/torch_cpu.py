"""ORACLE — TEST INFRASTRUCTURE ONLY.  The same path as clip_oracle.py (reference model/base/model.py:153-252, :359-372 and
model/modelbase.py:25-35) written with plain PyTorch CPU ops — `F.conv2d`, `F.multi_head_attention_forward`, `F.layer_norm`,
the calls the reference's own modules make — so that bench.py's cpu_baseline can time what the reference's CPU path costs
with ATen's threaded GEMMs (SURVEY 8d asks for both the restatement and this).  Checked against clip_oracle in
tests/test_oracle_clip.py."""
import torch
import torch.nn.functional as F


def _t(sd):
    return {k: torch.from_numpy(v) for k, v in sd.items()}


def _block(x, sd, p, heads, mask):
    d = x.shape[-1]
    h = F.layer_norm(x, (d,), sd[p + "ln_1.weight"], sd[p + "ln_1.bias"], 1e-5)
    a = F.multi_head_attention_forward(h, h, h, d, heads, sd[p + "attn.in_proj_weight"], sd[p + "attn.in_proj_bias"], None, None, False,
                                       0.0, sd[p + "attn.out_proj.weight"], sd[p + "attn.out_proj.bias"], training=False,
                                       need_weights=False, attn_mask=mask)[0]
    x = x + a
    h = F.layer_norm(x, (d,), sd[p + "ln_2.weight"], sd[p + "ln_2.bias"], 1e-5)
    h = F.linear(h, sd[p + "mlp.c_fc.weight"], sd[p + "mlp.c_fc.bias"])
    h = h * torch.sigmoid(1.702 * h)
    return x + F.linear(h, sd[p + "mlp.c_proj.weight"], sd[p + "mlp.c_proj.bias"])


def _layers(sd, prefix):
    n = 0
    while f"{prefix}{n}.ln_1.weight" in sd:
        n += 1
    return n


class TorchClip:
    def __init__(self, state_dict):
        self.sd = _t(state_dict)

    @torch.no_grad()
    def encode_image(self, image):
        sd = self.sd
        x = F.conv2d(torch.as_tensor(image), sd["visual.conv1.weight"], stride=sd["visual.conv1.weight"].shape[-1])
        B, d = x.shape[0], x.shape[1]
        x = x.reshape(B, d, -1).permute(0, 2, 1)
        x = torch.cat([sd["visual.class_embedding"].expand(B, 1, d), x], 1) + sd["visual.positional_embedding"]
        x = F.layer_norm(x, (d,), sd["visual.ln_pre.weight"], sd["visual.ln_pre.bias"], 1e-5).permute(1, 0, 2)
        for i in range(_layers(sd, "visual.transformer.resblocks.")):
            x = _block(x, sd, f"visual.transformer.resblocks.{i}.", d // 64, None)
        x = F.layer_norm(x[0], (d,), sd["visual.ln_post.weight"], sd["visual.ln_post.bias"], 1e-5)
        return x @ sd["visual.proj"]

    @torch.no_grad()
    def encode_text(self, text):
        sd = self.sd
        text = torch.as_tensor(text)
        L = text.shape[1]
        x = sd["token_embedding.weight"][text] + sd["positional_embedding"][:L]
        d = x.shape[-1]
        mask = torch.full((L, L), float("-inf")).triu_(1)
        x = x.permute(1, 0, 2)
        for i in range(_layers(sd, "transformer.resblocks.")):
            x = _block(x, sd, f"transformer.resblocks.{i}.", d // 64, mask)
        x = F.layer_norm(x.permute(1, 0, 2), (d,), sd["ln_final.weight"], sd["ln_final.bias"], 1e-5)
        return x[torch.arange(x.shape[0]), text.argmax(-1)] @ sd["text_projection"]


@torch.no_grad()
def linear_hash_codes(feat, w, b):
    return torch.sign(torch.tanh(F.linear(feat, torch.as_tensor(w), torch.as_tensor(b))))
