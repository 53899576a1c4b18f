"""Recurrent cells with the reference's names and parameter layout
(reference model_convlstm.py:73-126 ConvLSTMCell, :238-295 ConvTWACell, :297-401 ConvTWA).
Parameter containers only; the step arithmetic is the EPI_TWA epilogue of
`uavsal_conv_gemm` driven by engine.py."""
import torch.nn as nn

from .model_feature import _no_eager


class ConvTWACell(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.height, self.width = input_size
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        self.rnn_conv = nn.Conv2d(input_dim + hidden_dim, hidden_dim, kernel_size, padding=self.padding, bias=bias)
        nn.init.kaiming_normal_(self.rnn_conv.weight, mode="fan_out")   # model_convlstm.py:274

    def forward(self, input_tensor, cur_state):
        _no_eager("ConvTWACell")


class ConvTWA(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, num_layers,
                 batch_first=False, bias=True, return_all_layers=False):
        super().__init__()
        if not isinstance(kernel_size, (tuple, list)):
            raise ValueError("`kernel_size` must be tuple or list of tuples")
        if num_layers != 1:
            raise NotImplementedError("the UAVSal path uses a single ConvTWA layer (model.py:328-329)")
        self.height, self.width = input_size
        self.input_dim, self.hidden_dim = input_dim, [hidden_dim]
        self.kernel_size, self.num_layers = [kernel_size], num_layers
        self.batch_first, self.bias, self.return_all_layers = batch_first, bias, return_all_layers
        self.cell_list = nn.ModuleList([ConvTWACell((self.height, self.width), input_dim, hidden_dim,
                                                    kernel_size, bias)])

    def forward(self, input_tensor, hidden_state=None):
        _no_eager("ConvTWA")


class ConvLSTMCell(nn.Module):
    """Parameter container of reference model_convlstm.py:73-130 (one conv (in+hid) -> 4*hid,
    gate order i, f, o, g).  The step runs as the EPI_LSTM epilogue of `uavsal_conv_gemm`."""

    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, bias):
        super().__init__()
        self.height, self.width = input_size
        self.input_dim, self.hidden_dim = input_dim, hidden_dim
        self.kernel_size = kernel_size
        self.padding = kernel_size[0] // 2, kernel_size[1] // 2
        self.bias = bias
        self.rnn_conv = nn.Conv2d(input_dim + hidden_dim, 4 * hidden_dim, kernel_size, padding=self.padding, bias=bias)
        nn.init.xavier_uniform_(self.rnn_conv.weight)          # model_convlstm.py:109

    def forward(self, input_tensor, cur_state):
        _no_eager("ConvLSTMCell")


class ConvLSTM(nn.Module):
    def __init__(self, input_size, input_dim, hidden_dim, kernel_size, num_layers,
                 batch_first=False, bias=True, return_all_layers=False):
        super().__init__()
        if not isinstance(kernel_size, (tuple, list)):
            raise ValueError("`kernel_size` must be tuple or list of tuples")
        if num_layers != 1:
            raise NotImplementedError("UAVSAL_LSTM uses a single ConvLSTM layer (model.py:1029)")
        self.height, self.width = input_size
        self.input_dim, self.hidden_dim = input_dim, [hidden_dim]
        self.kernel_size, self.num_layers = [kernel_size], num_layers
        self.batch_first, self.bias, self.return_all_layers = batch_first, bias, return_all_layers
        self.cell_list = nn.ModuleList([ConvLSTMCell((self.height, self.width), input_dim, hidden_dim,
                                                     kernel_size, bias)])

    def forward(self, input_tensor, hidden_state=None):
        _no_eager("ConvLSTM")
