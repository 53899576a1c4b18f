"""MI355X-native (gfx950) implementation of the UAVSal per-frame saliency inference path
of zhangkao/IIP_UAVSal_Saliency.  `UAVSal` is a drop-in for the reference's
`model.UAVSal` on that path; see DESIGN.md and INTEGRATION.md."""
from .model import UAVSal, UAVSAL_LSTM, BasicConv2d, dwBlock, STBlock, spConv, teConv_sub, uavsal_srfnet_aspp, init_weights  # noqa: F401
from .model_feature import ReMobileNetV2  # noqa: F401
from .model_convlstm import ConvTWA, ConvTWACell, ConvLSTM, ConvLSTMCell  # noqa: F401

__all__ = ["UAVSal", "UAVSAL_LSTM", "ReMobileNetV2", "ConvTWA", "ConvTWACell", "ConvLSTM", "ConvLSTMCell"]
