"""models.backbone.build_backbone -- same signature as the reference (backbone/__init__.py:4-10).

'resnet' is ResNet-50 exactly as the reference wires it; 'resnet101' is added for the headline
benchmark config (the reference defines ResNet101 but never selects it).  `pretrained` is honoured
for both and nothing is ever fetched from the network.
"""
from models.backbone import mobilenet, resnet


def build_backbone(backbone, output_stride, batchnorm, mc_dropout, input_channels, pretrained):
    if backbone == 'resnet':
        return resnet.ResNet50(output_stride, batchnorm, pretrained=pretrained)
    elif backbone == 'resnet101':
        return resnet.ResNet101(output_stride, batchnorm, pretrained=pretrained)
    elif backbone == 'mobilenet':
        return mobilenet.MobileNetV2(output_stride=output_stride, batchnorm=batchnorm, mc_dropout=mc_dropout,
                                     input_channels=input_channels, pretrained=pretrained)
    else:
        raise NotImplementedError
