"""oracle/vgg16.py — torch-CPU restatement of the VGGFace VGG-16 feature model.  TEST INFRASTRUCTURE.

PARITY UNPINNED by the reference: siamese.FaceVGG16 (reference code/siamese.py:187-200) wraps
keras_vggface.VGGFace(model='vgg16', include_top=False) cut at 'pool5' + Flatten and
utils.preprocess_input(version=1); keras-vggface==0.5 is not vendored.  Restated from the published
keras_vggface/models.py (VGG16) and utils.py."""
import numpy as np
import torch
import torch.nn.functional as F

BLOCKS = (2, 2, 3, 3, 3)
MEAN_BGR = (93.5940, 104.7624, 129.1863)


def preprocess_input_v1(x):
    x = np.ascontiguousarray(np.array(x, dtype=np.float32, copy=True)[..., ::-1])
    for c in range(3):
        x[..., c] -= MEAN_BGR[c]
    return x


def forward(params, x_pre):
    with torch.no_grad():
        x = torch.from_numpy(np.ascontiguousarray(x_pre)).float().permute(0, 3, 1, 2)
        for b in range(5):
            for l in range(BLOCKS[b]):
                n = "conv%d_%d" % (b + 1, l + 1)
                w = torch.from_numpy(params[n + "/kernel"]).permute(3, 2, 0, 1).contiguous()
                x = F.relu(F.conv2d(x, w, torch.from_numpy(params[n + "/bias"]), padding=1))
            x = F.max_pool2d(x, 2, 2)
        return x.permute(0, 2, 3, 1).reshape(x.shape[0], -1).numpy()      # Keras Flatten: (h, w, c)


def process(params, X):
    return forward(params, preprocess_input_v1(X))
