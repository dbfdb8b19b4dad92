"""Host-side image pipeline of the reference (`transforms.py`; used by train.py:37-47 / test.py:70-76 as
`T.Compose([T.Resize(args.img_size, args.img_size), T.ToTensor(), T.Normalize(mean, std)])`), without torchvision (absent from this image):
the same operations written on PIL / numpy.  torchvision's `F.resize` on a PIL image is `img.resize((w, h), BILINEAR)` (NEAREST for the
target), `F.to_tensor` is uint8 HWC -> float CHW / 255, `F.normalize` is (x - mean) / std per channel."""
import numpy as np
import torch
from PIL import Image


class Compose:
    def __init__(self, transforms):
        self.transforms = transforms

    def __call__(self, image, target):
        for t in self.transforms:
            image, target = t(image, target)
        return image, target


class Resize:
    """transforms.py:20-31: image bilinear, target nearest, both to exactly (h, w)"""

    def __init__(self, h, w):
        self.h, self.w = h, w

    def __call__(self, image, target):
        image = image.resize((self.w, self.h), Image.BILINEAR)
        if target is not None:
            target = target.resize((self.w, self.h), Image.NEAREST)
        return image, target


class ToTensor:
    """transforms.py:83-87"""

    def __call__(self, image, target):
        a = np.asarray(image)
        if a.ndim == 2:
            a = a[:, :, None]
        t = torch.from_numpy(np.ascontiguousarray(a.transpose(2, 0, 1)))
        image = t.to(torch.float32).div(255) if t.dtype == torch.uint8 else t.to(torch.float32)
        if target is not None:
            target = torch.as_tensor(np.asarray(target).copy(), dtype=torch.int64)
        return image, target


class Normalize:
    """transforms.py:106-113"""

    def __init__(self, mean, std):
        self.mean, self.std = torch.tensor(mean, dtype=torch.float32).view(-1, 1, 1), torch.tensor(std, dtype=torch.float32).view(-1, 1, 1)

    def __call__(self, image, target):
        return (image - self.mean) / self.std, target


def get_transform(img_size):
    """train.py:37-47 `get_transform(args)`"""
    return Compose([Resize(img_size, img_size), ToTensor(), Normalize(mean=[0.485, 0.456, 0.406], std=[0.229, 0.224, 0.225])])
