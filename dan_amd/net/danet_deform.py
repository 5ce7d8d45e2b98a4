"""DAN-Deform graph on the libdanhip kernels — mirrors net/danet_deform.py: DAN with the context module V2
(se_inception_block :267-290 = 1x1 down to 256 (ReLU) -> deformable 3x3, 4 deformable groups, bias -> ReLU -> 1x1 up (ReLU)
+ residual)."""
from ..utility import custom_op
from . import danet


class VGG16Backbone(danet.VGG16Backbone):
    def se_inception_block(self, inputs, name=None):
        """net/danet_deform.py:267-290."""
        c = inputs.shape[-1]
        d = self._cr(inputs, 256, (1, 1), name + "/conv_1x1_down")
        y = custom_op.deform_conv_2d(d, 256, 3, 3, stride=1, dilate_rate=1, deformable_group=4, data_format="channels_last", no_bias=False,
                                     name=name + "/deform_conv", variables=self.vs, relu=True)
        return self._residual(self._cr(y, c, (1, 1), name + "/conv_1x1_up"), inputs)
