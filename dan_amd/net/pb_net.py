"""PyramidBox graph on the libdanhip kernels — mirrors net/pb_net.py (class VGG16Backbone: the S3FD backbone plus
build_lfpn :185-226, context_pred_module :158-183, get_predict_module :230-290) with the reference's variable names."""
import torch

from .. import ops
from . import sfd_net


class VGG16Backbone(sfd_net.VGG16Backbone):
    def build_lfpn(self, feature_layers, skip_last=3, name=None, fused_channels=None):
        """net/pb_net.py:185-226.  No activation on any of the three convs; the running `up_sampling` is the SUM
        (lateral + upsampled), not the fused output.  (fused_channels: DAN's variant fixes the fused conv to 256.)"""
        name = name or "lfpn"
        output_layers = []
        up_sampling = None
        for ind in range(skip_last, 0, -1):
            sc = "{}/fpn_{}".format(name, ind - 1)
            down_channels = feature_layers[ind - 1].shape[-1]
            if up_sampling is None:
                up_sampling = feature_layers[ind]
            up_sampling = self.conv2d(up_sampling, down_channels, (1, 1), 1, sc + "/upsample_conv", relu=False)
            lateral = self.conv2d(feature_layers[ind - 1], down_channels, (1, 1), 1, sc + "/lateral", relu=False)
            up_sampling = ops.resize_bilinear_add(up_sampling, lateral)         # lateral + resize_bilinear(up, size(lateral))
            featmap = self.conv2d(up_sampling, fused_channels or down_channels, (3, 3), 1, sc + "/fused_conv", relu=False)
            output_layers.append(featmap)
        return list(reversed(output_layers)) + list(feature_layers[skip_last:])

    def context_pred_module(self, feature_layers):
        """net/pb_net.py:158-183 (hard-coded 1024 channels)."""
        def block(inputs, num_channels, last_div, name):
            inputs = self.conv_relu(inputs, num_channels, (3, 3), (1, 1), name + "/conv1")
            inputs = self.conv_relu(inputs, num_channels // 4, (3, 3), (1, 1), name + "/conv2")
            return self.conv_relu(inputs, num_channels // last_div, (3, 3), (1, 1), name + "/conv3")
        output_layers = []
        for ind, featmap in enumerate(feature_layers):
            nc = 1024
            branch1 = block(featmap, nc, 4, "cpm/branch{}_1".format(ind))
            branch2 = block(featmap, nc, 4, "cpm/branch{}_2".format(ind))
            branch2_1 = block(branch2, nc, 8, "cpm/branch{}_2_1".format(ind))
            branch2_2_1 = block(branch2, nc, 8, "cpm/branch{}_2_2_1".format(ind))
            branch2_2_2 = block(branch2_2_1, nc, 8, "cpm/branch{}_2_2_2".format(ind))
            output_layers.append(ops.concat([branch1, branch2_1, branch2_2_2]))
        return output_layers

    def get_predict_module(self, feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer, name=None):
        """net/pb_net.py:230-290: returns (location_pred [B,A,4], cls_pred [B,A,2]) over the given levels."""
        return self.predict_heads(feature_layers, pos_maxout, neg_maxout, num_anchors_depth_per_layer, name or "predict_face")
