"""readMTP — drop-in for the parts of reference code/readMTP.py the Multi-PIE driver calls:
qualifies / readAllImages (code/readMTP.py:8-39), resizeImages (:116-119), getGenerator (:80-113) and
createMiniBatch (:123-135).  Image files are read with PIL; resizing is the device bilinear kernel."""
import os

import numpy as np

from .alink_loop import createMiniBatchMTP as createMiniBatch  # noqa: F401
from .pairs import getGeneratorMTP as getGenerator, getNormalGenerator  # noqa: F401

_SUFFIXES = ("01_01_051_06.png", "02_01_051_06.png", "01_01_051_08.png", "02_01_051_08.png")


def qualifies(path):
    """the four (session, recording, camera 05_1, illumination 06/08) Multi-PIE shots the paper uses"""
    return path.endswith(_SUFFIXES)


def resizeImages(images, resize_res):
    from . import noise as _noise
    return [np.asarray(_noise.resize_images(images[0], resize_res)), np.asarray(_noise.resize_images(images[1], resize_res))]


def readAllImages(dirPath, resize=None):
    """-> list (one entry per person id = the file-name prefix before the first '_') of (k, H, W, C) arrays"""
    from PIL import Image
    from . import noise as _noise
    person_wise = {}
    for path in os.listdir(dirPath):
        if qualifies(path):
            person_wise.setdefault(int(path.split('_')[0]), []).append(path)
    people = []
    for key in person_wise:
        imgs = []
        for name in person_wise[key]:
            img = np.asarray(Image.open(os.path.join(dirPath, name)), dtype=np.float32)
            if resize:
                img = np.asarray(_noise.resize_images(img[None], resize))[0]
            imgs.append(img)
        people.append(np.stack(imgs))
    return people
