"""Static description of the MonoCon (DLA-34 + DLAUp + dense heads) parameter set.

This is the single host-side source of truth for *names and shapes*: the product
modules (``model/``), the synthetic generator (``synth.py``) and the C-ABI binder
(``engine.py``) all walk the same table, so the 449-entry ``state_dict`` keeps the
reference's key names and OIHW layout for checkpoint interchange
(reference: model/backbone/dla.py:208-298, model/backbone/dla_neck.py:41-130,
model/dense_heads/monocon_heads.py:76-131, model/norm/attentive_norm.py:118-148).

Entry kinds
    conv    : ``<name>.weight`` (O, I, kh, kw) [+ ``<name>.bias`` (O,)]
    bn      : weight, bias, running_mean, running_var (C,), num_batches_tracked ()
    deconv  : ``<name>.weight`` (C, 1, 4, 4)   depthwise ConvTranspose2d
    attnbn  : weight_, bias_ (10, C), running_mean, running_var, num_batches_tracked,
              attn_weights.attention.0.weight (10, C, 1, 1), attn_weights.attention.1.<bn>
"""
from collections import OrderedDict

DLA34_LEVELS = (1, 1, 1, 2, 2, 1)
DLA34_CHANNELS = (16, 32, 64, 128, 256, 512)
NUM_AFFINE = 10

# (attribute name, output channels, has its own 1x1 output conv at index 3)
HEAD_BRANCHES = (
    ("heatmap_head", 3),
    ("wh_head", 2),
    ("offset_head", 2),
    ("center2kpt_offset_head", 18),
    ("kpt_heatmap_head", 9),
    ("kpt_heatmap_offset_head", 2),
    ("dim_head", 3),
    ("depth_head", 2),
)

# prediction-dict key -> (producing branch, channels); order = pred_dict order of the
# reference (monocon_heads.py:190-200)
PRED_KEYS = (
    ("center_heatmap_pred", 3),
    ("kpt_heatmap_pred", 9),
    ("wh_pred", 2),
    ("offset_pred", 2),
    ("kpt_heatmap_offset_pred", 2),
    ("center2kpt_offset_pred", 18),
    ("dim_pred", 3),
    ("depth_pred", 2),
    ("alpha_cls_pred", 12),
    ("alpha_offset_pred", 12),
)

LOSS_KEYS = (
    "loss_center_heatmap", "loss_wh", "loss_offset", "loss_dim",
    "loss_center2kpt_offset", "loss_kpt_heatmap", "loss_kpt_heatmap_offset",
    "loss_alpha_cls", "loss_alpha_reg", "loss_depth",
)


class _Spec:
    def __init__(self):
        self.entries = []          # (kind, name, meta)

    def conv(self, name, cin, cout, k, bias=False):
        self.entries.append(("conv", name, dict(cin=cin, cout=cout, k=k, bias=bias)))

    def bn(self, name, c):
        self.entries.append(("bn", name, dict(c=c)))

    def deconv(self, name, c):
        self.entries.append(("deconv", name, dict(c=c)))

    def attnbn(self, name, c):
        self.entries.append(("attnbn", name, dict(c=c)))


def _basic_block(s, name, cin, cout):
    s.conv(name + ".conv1", cin, cout, 3)
    s.bn(name + ".bn1", cout)
    s.conv(name + ".conv2", cout, cout, 3)
    s.bn(name + ".bn2", cout)


def _tree(s, name, levels, cin, cout, level_root, root_dim=0):
    """Registration order follows nn.Module attribute order of the reference Tree:
    tree1, tree2, root (levels==1 only), project."""
    if root_dim == 0:
        root_dim = 2 * cout
    if level_root:
        root_dim += cin
    if levels == 1:
        _basic_block(s, name + ".tree1", cin, cout)
        _basic_block(s, name + ".tree2", cout, cout)
        s.conv(name + ".root.conv", root_dim, cout, 1)
        s.bn(name + ".root.bn", cout)
    else:
        _tree(s, name + ".tree1", levels - 1, cin, cout, False, 0)
        _tree(s, name + ".tree2", levels - 1, cout, cout, False, root_dim + cout)
    if cin != cout:
        s.conv(name + ".project.0", cin, cout, 1)
        s.bn(name + ".project.1", cout)


def build_spec():
    s = _Spec()
    ch = DLA34_CHANNELS
    s.conv("backbone.base_layer.0", 3, ch[0], 7)
    s.bn("backbone.base_layer.1", ch[0])
    s.conv("backbone.level0.0", ch[0], ch[0], 3)
    s.bn("backbone.level0.1", ch[0])
    s.conv("backbone.level1.0", ch[0], ch[1], 3)
    s.bn("backbone.level1.1", ch[1])
    _tree(s, "backbone.level2", DLA34_LEVELS[2], ch[1], ch[2], False)
    _tree(s, "backbone.level3", DLA34_LEVELS[3], ch[2], ch[3], True)
    _tree(s, "backbone.level4", DLA34_LEVELS[4], ch[3], ch[4], True)
    _tree(s, "backbone.level5", DLA34_LEVELS[5], ch[4], ch[5], True)

    # DLAUp over levels 2..5: ida_0 ([256,512]->256), ida_1 ([128,256,256]->128),
    # ida_2 ([64,128,128,128]->64)   (dla_neck.py:121-128)
    neck_in = [64, 128, 256, 512]
    for i in range(3):
        j = -i - 2
        ins = list(neck_in[j:])
        out = neck_in[j]
        for t in range(1, len(ins)):
            pre = "neck.ida_%d." % i
            s.conv(pre + "proj_%d.conv" % t, ins[t], out, 3)
            s.bn(pre + "proj_%d.bn1" % t, out)
            s.deconv(pre + "up_%d" % t, out)
            s.conv(pre + "node_%d.conv" % t, 2 * out, out, 3)
            s.bn(pre + "node_%d.bn1" % t, out)
        for t in range(len(neck_in) + j + 1, len(neck_in)):
            neck_in[t] = out

    for hname, cout in HEAD_BRANCHES:
        s.conv("head.%s.0" % hname, 64, 64, 3, bias=True)
        s.attnbn("head.%s.1" % hname, 64)
        s.conv("head.%s.3" % hname, 64, cout, 1, bias=True)
    s.conv("head.dir_feat.0", 64, 64, 3, bias=True)
    s.attnbn("head.dir_feat.1", 64)
    s.conv("head.dir_cls.0", 64, 12, 1, bias=True)
    s.conv("head.dir_reg.0", 64, 12, 1, bias=True)
    return s


def _bn_fields(name, c):
    return [
        (name + ".weight", (c,), "f32", "param"),
        (name + ".bias", (c,), "f32", "param"),
        (name + ".running_mean", (c,), "f32", "buffer"),
        (name + ".running_var", (c,), "f32", "buffer"),
        (name + ".num_batches_tracked", (), "i64", "buffer"),
    ]


def state_fields(spec=None):
    """Flat list of (key, shape, dtype, role) in the reference's state_dict order."""
    spec = spec or build_spec()
    out = []
    for kind, name, m in spec.entries:
        if kind == "conv":
            out.append((name + ".weight", (m["cout"], m["cin"], m["k"], m["k"]), "f32", "param"))
            if m["bias"]:
                out.append((name + ".bias", (m["cout"],), "f32", "param"))
        elif kind == "bn":
            out += _bn_fields(name, m["c"])
        elif kind == "deconv":
            out.append((name + ".weight", (m["c"], 1, 4, 4), "f32", "param"))
        elif kind == "attnbn":
            c = m["c"]
            out.append((name + ".weight_", (NUM_AFFINE, c), "f32", "param"))
            out.append((name + ".bias_", (NUM_AFFINE, c), "f32", "param"))
            out.append((name + ".running_mean", (c,), "f32", "buffer"))
            out.append((name + ".running_var", (c,), "f32", "buffer"))
            out.append((name + ".num_batches_tracked", (), "i64", "buffer"))
            out.append((name + ".attn_weights.attention.0.weight", (NUM_AFFINE, c, 1, 1), "f32", "param"))
            out += _bn_fields(name + ".attn_weights.attention.1", NUM_AFFINE)
    return out


def state_shapes():
    return OrderedDict((k, (shape, dt, role)) for k, shape, dt, role in state_fields())


# parameters the reference never routes a gradient to (outer ``project`` of the
# two-level trees; dla.py:193-194 recomputes the residual inside the nested Tree)
DEAD_PARAMS = tuple(
    "backbone.level%d.project.%s" % (lv, f)
    for lv in (3, 4)
    for f in ("0.weight", "1.weight", "1.bias")
)
