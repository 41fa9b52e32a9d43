"""oracle/vgg_resnet50.py — torch-CPU restatement of the VGGFace2 ResNet-50 feature model.

TEST INFRASTRUCTURE (see oracle/__init__.py).  PARITY UNPINNED by the reference: siamese.RESNET50
(reference code/siamese.py:203-216) wraps keras_vggface.VGGFace(model='resnet50', include_top=False)
cut at 'avg_pool' + Flatten and keras_vggface.utils.preprocess_input(version=2); keras-vggface==0.5
(reference requirements.txt:19) and its downloaded weights are not vendored and cannot be installed.
Restated from the published keras_vggface/models.py (RESNET50, resnet_conv_block,
resnet_identity_block) and utils.py (preprocess_input), unfused and layer by layer:
  conv 7x7/2 'same' -> BN(eps 1e-3) -> ReLU -> MaxPool 3x3/2 'valid' -> bottleneck stages
  [3,4,6,3] (stride on the first 1x1 of conv3_1/conv4_1/conv5_1) -> AveragePooling2D((7,7)) -> Flatten.
"""
import numpy as np
import torch
import torch.nn.functional as F

UNITS = (3, 4, 6, 3)
MID = (64, 128, 256, 512)
BN_EPS = 1e-3
MEAN_BGR = (91.4953, 103.8827, 131.0912)


def tensor_shapes():
    """Ordered {name: shape} in Keras layouts: kernel (kh, kw, in, out), BN vectors (out,)."""
    t = {}

    def conv(name, k, cin, cout):
        t[name + "/kernel"] = (k, k, cin, cout)
        for s in ("gamma", "beta", "moving_mean", "moving_variance"):
            t[name + "/bn/" + s] = (cout,)
    conv("conv1/7x7_s2", 7, 3, 64)
    cin = 64
    for s in range(4):
        mid, out = MID[s], 4 * MID[s]
        for u in range(1, UNITS[s] + 1):
            p = "conv%d_%d_" % (s + 2, u)
            conv(p + "1x1_reduce", 1, cin, mid)
            conv(p + "3x3", 3, mid, mid)
            conv(p + "1x1_increase", 1, mid, out)
            if u == 1:
                conv(p + "1x1_proj", 1, cin, out)
            cin = out
    return t


def preprocess_input_v2(x):
    """keras_vggface.utils.preprocess_input(x, version=2), channels_last: RGB -> BGR, subtract means."""
    x = np.array(x, dtype=np.float32, copy=True)[..., ::-1]
    x = np.ascontiguousarray(x)
    x[..., 0] -= MEAN_BGR[0]
    x[..., 1] -= MEAN_BGR[1]
    x[..., 2] -= MEAN_BGR[2]
    return x


def flops_per_image(size=(224, 224)):
    h, w = (size[0] + 1) // 2, (size[1] + 1) // 2
    macs = h * w * 64 * 147
    h, w = (h - 3) // 2 + 1, (w - 3) // 2 + 1
    cin = 64
    for s in range(4):
        mid, out = MID[s], 4 * MID[s]
        for u in range(1, UNITS[s] + 1):
            st = 2 if (u == 1 and s > 0) else 1
            ho, wo = (h - 1) // st + 1, (w - 1) // st + 1
            macs += ho * wo * (cin * mid + 9 * mid * mid + mid * out + (cin * out if u == 1 else 0))
            h, w, cin = ho, wo, out
    return 2.0 * macs


def forward(params, x_pre, dtype=torch.float32, threads=None):
    """x_pre: (N, H, W, 3) already preprocessed (BGR, mean-subtracted).  -> (N, 2048) numpy."""
    if threads:
        torch.set_num_threads(threads)
    t = lambda n: torch.from_numpy(np.ascontiguousarray(params[n])).to(dtype)

    def conv_bn(x, name, stride=1, pad=0, relu=True):
        w = t(name + "/kernel").permute(3, 2, 0, 1).contiguous()
        x = F.conv2d(x, w, stride=stride, padding=pad)
        g, b, mu, var = (t(name + "/bn/" + s) for s in ("gamma", "beta", "moving_mean", "moving_variance"))
        x = (x - mu[None, :, None, None]) / torch.sqrt(var[None, :, None, None] + BN_EPS) * g[None, :, None, None] \
            + b[None, :, None, None]
        return F.relu(x) if relu else x
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(x_pre)).to(dtype).permute(0, 3, 1, 2)
        H, W = x.shape[2], x.shape[3]
        th = max((-(-H // 2) - 1) * 2 + 7 - H, 0)
        tw = max((-(-W // 2) - 1) * 2 + 7 - W, 0)
        x = F.pad(x, (tw // 2, tw - tw // 2, th // 2, th - th // 2))          # TensorFlow 'same'
        x = conv_bn(x, "conv1/7x7_s2", stride=2)
        x = F.max_pool2d(x, 3, 2)
        for s in range(4):
            for u in range(1, UNITS[s] + 1):
                p = "conv%d_%d_" % (s + 2, u)
                st = 2 if (u == 1 and s > 0) else 1
                y = conv_bn(x, p + "1x1_reduce", stride=st)
                y = conv_bn(y, p + "3x3", pad=1)
                y = conv_bn(y, p + "1x1_increase", relu=False)
                sc = conv_bn(x, p + "1x1_proj", stride=st, relu=False) if u == 1 else x
                x = F.relu(y + sc)
        x = F.avg_pool2d(x, 7)
        out = x.permute(0, 2, 3, 1).reshape(x.shape[0], -1).numpy()
        return out.astype(np.float32 if dtype == torch.float32 else np.float64)


def process(params, X, **kw):
    """RESNET50.process (code/siamese.py:215-216): predict(preprocess(X))."""
    return forward(params, preprocess_input_v2(X), **kw)
