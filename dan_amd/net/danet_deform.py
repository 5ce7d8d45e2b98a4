"""DAN-Deform graph on the libdanhip kernels — mirrors net/danet_deform.py: DAN with the context module V2
(se_inception_block :267-290 = 1x1 down to 256 (ReLU) -> deformable 3x3, 4 deformable groups, bias -> ReLU -> 1x1 up (ReLU)
+ residual)."""
import torch

from .. import ops
from ..utility import custom_op
from . import danet


class VGG16Backbone(danet.VGG16Backbone):
    def se_inception_block(self, inputs, name=None):
        """net/danet_deform.py:267-290."""
        c = inputs.shape[-1]
        d = self._cr(inputs, 256, (1, 1), name + "/conv_1x1_down")
        y = custom_op.deform_conv_2d(d, 256, 3, 3, stride=1, dilate_rate=1, deformable_group=4, data_format="channels_last", no_bias=False,
                                     name=name + "/deform_conv", variables=self.vs, relu=True)
        return self._residual_conv(y, c, name + "/conv_1x1_up", inputs)

    def _deform_relu(self, feat, name):
        return custom_op.deform_conv_2d(feat, feat.shape[-1], 3, 3, stride=1, dilate_rate=1, deformable_group=4, data_format="channels_last",
                                        no_bias=False, name=name, variables=self.vs, relu=True)

    def get_features_stage1_conv_only(self, feature_layers, name=None):
        """net/danet_deform.py:328-339 (unused variant): one deformable 3x3 conv + ReLU per level, channels kept."""
        name = name or "prediction_modules_stage1"
        return [self._deform_relu(f, "{}/predict_stage1_conv{}".format(name, i)) for i, f in enumerate(feature_layers)]

    def get_features_stage2_conv_only(self, feature_stage1, feature_layers, name=None):
        """net/danet_deform.py:341-366 (unused variant): the stage-2 input mix (stop_gradient(stage 1) -> 1x1 C//3, feature -> 1x1
        C - C//3, concat) followed by one deformable 3x3 conv + ReLU."""
        name = name or "prediction_modules_stage2"
        outs = []
        for i, f in enumerate(feature_layers):
            c = f.shape[-1]
            s1 = self._cr(ops.stop_gradient(feature_stage1[i]), c // 3, (1, 1), "{}/satge1_conv_1x1_{}".format(name, i))            # (sic)
            rs = self._cr(f, c - c // 3, (1, 1), "{}/residual_conv_1x1_{}".format(name, i))
            outs.append(self._deform_relu(ops.concat([s1, rs]), "{}/predict_stage2_conv{}".format(name, i)))
        return outs
