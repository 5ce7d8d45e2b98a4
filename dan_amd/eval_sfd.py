"""eval_sfd.py's test-time pipeline (:60-196, :323-330): the same helpers as eval_dan.py without the pyramid pass."""
from .eval_dan import (Detector, bbox_vote, bbox_vote_batch, detect_face, flip_test, get_shrink, multi_scale_test,  # noqa: F401
                       resize_image, write_to_txt)
from .eval_dan import detect_image as _detect_image


def detect_image(net, image):
    """eval_sfd.py:323-328: origin + flip + multi-scale, merged by box voting."""
    return _detect_image(net, image, pyramid=False)


def detect_images(net, images):
    """The same passes for B images of one size, batched (eval_dan.detect_images): (dets [B, 750, 5], num [B]) on the device."""
    from .eval_dan import detect_images as _detect_images
    return _detect_images(net, images, pyramid=False)
