"""ResNet-50/101/152 + batch-norm backbone of the reference's net/resnet_danet.py (class ResNetBackbone :92-228; imported by no
script — SURVEY §8f row 4) on the libdanhip kernels.  The LFPN / context-module / prediction-head methods of that class are the
DAN ones (same code in the reference), inherited here from dan_amd.net.danet.VGG16Backbone."""
import torch

from .. import ops
from . import danet


class ResNetBackbone(danet.VGG16Backbone):
    _block_settings = {50: (3, 4, 6, 3), 101: (3, 4, 23, 3), 152: (3, 8, 36, 3)}                       # :106-110

    def __init__(self, depth=50, data_format="channels_last", freeze_bn=False, bn_epsilon=1e-05, bn_momentum=0.997, use_fused_bn=True,
                 variables=None):
        super().__init__(data_format, bn_epsilon, bn_momentum, use_fused_bn, variables=variables)
        if depth not in self._block_settings:
            raise ValueError("depth must be 50, 101 or 152")
        self._depth = depth
        self._bn_trainable = not freeze_bn

    # conv (no bias, 'same' or 'valid') -> batch norm -> optional ReLU; batch statistics only when training and BN is trainable (:183-184)
    def _cbn(self, inputs, filters, kernel_size, strides, scope, training, padding, relu):
        kh, kw = kernel_size
        w = self.vs.get(scope + "/conv2d/kernel", (kh, kw, getattr(inputs, "_real_channels", inputs.shape[-1]), filters), "glorot")
        y = ops.conv2d(inputs, w, None, stride=strides, relu=False, padding=padding)
        return self._bn(y, scope, training and self._bn_trainable, relu=relu)

    def conv_bn_relu(self, inputs, filters, kernel_size, strides, scope, training, padding="same", dilate_rate=1, reuse=None):
        """net/resnet_danet.py:176-191."""
        assert dilate_rate == 1
        return self._cbn(inputs, filters, kernel_size, strides[0] if isinstance(strides, (tuple, list)) else strides, scope, training, padding, True)

    def conv_bn(self, inputs, filters, kernel_size, strides, scope, training, padding="same", reuse=None):
        """net/resnet_danet.py:203-218."""
        return self._cbn(inputs, filters, kernel_size, strides[0] if isinstance(strides, (tuple, list)) else strides, scope, training, padding, False)

    def bottleneck_block(self, inputs, filters, scope, training, need_reduce=True, is_root=False, reuse=None):
        """net/resnet_danet.py:157-173: 1x1 reduce (stride here) -> 3x3 -> 1x1 increase, projection shortcut on the first unit."""
        strides = 1 if (not need_reduce) or is_root else 2
        shortcut = self.conv_bn(inputs, filters * 2, (1, 1), strides, scope + "/shortcut", training, padding="valid") if need_reduce else inputs
        y = self.conv_bn_relu(inputs, filters // 2, (1, 1), strides, scope + "/reduce", training, padding="valid")
        # the reference pads by one pixel and convolves 'valid': identical to the 3x3 'same' convolution at stride 1
        y = self.conv_bn_relu(y, filters // 2, (3, 3), 1, scope + "/block_3x3", training, padding="same")
        y = self.conv_bn(y, filters * 2, (1, 1), 1, scope + "/increase", training)
        return torch.relu(y + shortcut)

    def get_featmaps(self, inputs, training=False, freeze=False):
        """net/resnet_danet.py:114-155.  inputs: the padded-to-8-channel BGR image tensor of sfd_net.prepare_input."""
        input_depth = [128, 256, 512, 1024]
        training_sts = training
        if freeze:
            training = False
        x = torch.nn.functional.pad(inputs, (0, 0, 3, 3, 3, 3))                                        # tf.pad by 3, then 7x7/2 'valid'
        x._real_channels = getattr(inputs, "_real_channels", inputs.shape[-1])
        x = self.conv_bn_relu(x, input_depth[0] // 2, (7, 7), 2, "block_0/conv_1", training, padding="valid")
        x = ops.max_pool_3x3_s2(x)
        collected, is_root = [], True
        for ind, num_unit in enumerate(self._block_settings[self._depth]):
            need_reduce = True
            for unit in range(1, num_unit + 1):
                x = self.bottleneck_block(x, input_depth[ind], "block_{}/conv_{}".format(ind + 1, unit), training, need_reduce, is_root)
                need_reduce, is_root = False, False
            if freeze and ind == 0:
                x = ops.stop_gradient(x)                                                                      # tf.stop_gradient: un-freeze from here
                training = training_sts
            collected.append(x)
        x = self.conv_bn_relu(x, 512, (1, 1), 1, "additional_layers/conv6_1", training)
        x = self.conv_bn_relu(x, 512, (3, 3), 2, "additional_layers/conv6_2", training)
        collected.append(x)
        x = self.conv_bn_relu(x, 128, (1, 1), 1, "additional_layers/conv7_1", training)
        x = self.conv_bn_relu(x, 256, (3, 3), 2, "additional_layers/conv7_2", training)
        collected.append(x)
        return collected
