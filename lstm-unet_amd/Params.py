"""Run configuration objects exposing the field names and default values of the reference's Params.py
(CTCParams :27-156, CTCInferenceParams :159-195) -- `params.<field>` is what train2D.py / Inference2D.py read.

Defaults live in two tables below and are installed as class attributes, so `CTCParams.batch_size` etc. work
exactly as upstream; a dict of command-line overrides is applied per instance (unknown keys only warn, as in
the reference's `_override_params_`, Params.py:17-23).

Differences on purpose: the default data provider is the synthetic clip stream (no dataset ships here); the
Cell-Tracking-Challenge RAM reader is selected with `data_provider='ctc'` / `--data_provider ctc`; `--root_data_dir` is honoured (upstream ignores it, Params.py:104-107);
only rank 0 creates output directories under data-parallel launches; `sync_bn` is new.
"""
import os
from datetime import datetime

import DataHandeling
import Networks as Nets

ROOT_DATA_DIR = '~/CellTrackingChallenge/Training/'
ROOT_TEST_DATA_DIR = '~/CellTrackingChallenge/Test/'
ROOT_SAVE_DIR = '~/LSTM-UNet-Outputs/'

_ENCODER_WIDTHS = (128, 256, 256, 512)
_DECODER_WIDTHS = (256, 128, 64, 32)


def lstm_unet_kernels(k_conv=3, k_lstm=5, n_classes=3):
    """The train2D.py default architecture (Params.py:49-69): k_conv x k_conv encoder / decoder convolutions,
    k_lstm x k_lstm ConvLSTM kernels, a final 1x1 convolution to the class logits."""
    up = [[(k_conv, w), (k_conv, w)] for w in _DECODER_WIDTHS]
    up[-1].append((1, n_classes))
    return {'down_conv_kernels': [[(k_conv, w), (k_conv, w)] for w in _ENCODER_WIDTHS],
            'lstm_kernels': [[(k_lstm, w)] for w in _ENCODER_WIDTHS],
            'up_conv_kernels': up}


_SIM = ('Fluo-N2DH-SIM+', '01'), ('Fluo-N2DH-SIM+', '02')

_TRAIN_DEFAULTS = dict(
    # general
    experiment_name='MyRun_SIM', gpu_id=0,
    # data
    data_provider_class=DataHandeling.SyntheticSequence2D, root_data_dir=ROOT_DATA_DIR,
    train_sequence_list=list(_SIM), val_sequence_list=list(_SIM),
    crop_size=(128, 128), batch_size=5, unroll_len=4, data_format='NCHW',
    train_q_capacity=200, val_q_capacity=200, num_val_threads=2, num_train_threads=8,
    # network
    net_model=Nets.ULSTMnet2D, net_kernel_params=lstm_unet_kernels(3, 5),
    # optimisation
    class_weights=[0.15, 0.25, 0.6], learning_rate=1e-5, num_iterations=1000000,
    validation_interval=1000, print_to_console_interval=10,
    # checkpoints
    load_checkpoint=False, load_checkpoint_path='', continue_run=False, save_checkpoint_dir=ROOT_SAVE_DIR,
    save_checkpoint_iteration=5000, save_checkpoint_every_N_hours=24, save_checkpoint_max_to_keep=5,
    # logging
    tb_sub_folder='LSTMUNet', write_to_tb_interval=500, save_log_dir=ROOT_SAVE_DIR,
    # debugging
    dry_run=False, profile=False,
    # MI355X options: pool BatchNorm statistics over all data-parallel ranks; MFMA operand precision
    sync_bn=False, precision='fp32', data_provider=None,
    resize='tf2.0',        # bilinear convention of the up blocks (Networks.py:143): 'tf2.0' = TensorFlow 2.0 / 2.1, 'half_pixel' = later
)

_INFER_DEFAULTS = dict(
    gpu_id=0, model_path='./Models/LSTMUNet2D/PhC-C2DL-PSC/', output_path='./tmp/output/PhC-C2DL-PSC/01',
    sequence_path=os.path.join(ROOT_TEST_DATA_DIR, 'PhC-C2DL-PSC/01/'), filename_format='t*.tif',
    data_reader=DataHandeling.CTCInferenceReader, data_format='NCHW',
    FOV=0, min_cell_size=10, max_cell_size=100, edge_dist=2, pre_sequence_frames=4,
    dry_run=False, save_intermediate=True, save_intermediate_path='./tmp/output/PhC-C2DL-PSC/01',
    precision='fp32',      # MI355X option: 'bf16' = bf16 MFMA operands; 'bf16x3' = fp32 arithmetic on the bf16 MFMA (exact 3-way split)
    resize=None,           # bilinear convention: None = what model_params.pickle recorded (else 'tf2.0'); 'tf2.0' / 'half_pixel' override
    fov_fix=False,         # MI355X option: True masks columns [0, FOV) instead of the reference's single column (Inference2D.py:97)
    graph=False,           # MI355X option: True replays the per-frame launch sequence from a captured hipGraph
)


class ParamsBase(object):
    aws = False

    def _override_params_(self, params_dict: dict):
        known = {name for klass in type(self).__mro__ for name in vars(klass)}
        for key, val in params_dict.items():
            if key not in known:
                print('Warning!: Parameter:{} not in defualt parameters'.format(key))
            setattr(self, key, val)


def _stamp():
    return datetime.now().strftime('%Y-%m-%d_%H%M%S')


class CTCParams(ParamsBase):
    """Training configuration; constructs the two clip providers and the run directories."""

    def __init__(self, params_dict):
        self._override_params_(params_dict)
        if getattr(self, 'data_provider', None) == 'ctc':       # --data_provider ctc: the reference's provider
            self.data_provider_class = DataHandeling.CTCRAMReaderSequence2D
        elif getattr(self, 'data_provider', None) == 'synthetic':
            self.data_provider_class = DataHandeling.SyntheticSequence2D
        root = os.path.expanduser(self.root_data_dir)
        self.train_data_base_folders = [(os.path.join(root, name), seq) for name, seq in self.train_sequence_list]
        self.val_data_base_folders = [(os.path.join(root, name), seq) for name, seq in self.val_sequence_list]
        shared = dict(image_crop_size=self.crop_size, unroll_len=self.unroll_len, deal_with_end=0,
                      batch_size=self.batch_size, data_format=self.data_format, randomize=True, return_dist=False,
                      queue_capacity=self.train_q_capacity,
                      rank=int(os.environ.get('RANK', '0')))      # data parallel: every rank streams its own clips
        self.train_data_provider = self.data_provider_class(sequence_folder_list=self.train_data_base_folders,
                                                            num_threads=self.num_train_threads, **shared)
        self.val_data_provider = self.data_provider_class(sequence_folder_list=self.val_data_base_folders,
                                                          num_threads=self.num_val_threads, **shared)
        self._resolve_run_dirs()
        self.channel_axis = 1 if self.data_format == 'NCHW' else 3

    def _resolve_run_dirs(self):
        if self.load_checkpoint and self.continue_run:
            # keep writing into the run that produced the checkpoint
            path = self.load_checkpoint_path.rstrip('/')
            base = path if os.path.isdir(path) else os.path.dirname(os.path.dirname(path))
            if os.path.basename(base) in ('tf-ckpt', 'tf_ckpts'):
                base = os.path.dirname(base)
            self.experiment_log_dir = self.experiment_save_dir = base
        else:
            leaf = os.path.join(self.tb_sub_folder, self.experiment_name, _stamp())
            self.experiment_log_dir = os.path.join(os.path.expanduser(self.save_log_dir), leaf)
            self.experiment_save_dir = os.path.join(os.path.expanduser(self.save_checkpoint_dir), leaf)
        if self.dry_run or int(os.environ.get('RANK', '0')) != 0:
            return
        for d in (self.experiment_save_dir, os.path.join(self.experiment_log_dir, 'train'),
                  os.path.join(self.experiment_log_dir, 'val')):
            os.makedirs(d, exist_ok=True)


class CTCInferenceParams(ParamsBase):
    """Streaming-inference configuration (Inference2D.py)."""

    def __init__(self, params_dict: dict = None):
        self._override_params_(params_dict or {})
        self.channel_axis = 1 if self.data_format == 'NCHW' else 3
        if self.dry_run:
            return
        os.makedirs(self.output_path, exist_ok=True)
        if self.save_intermediate:
            top = os.path.join(self.save_intermediate_path, 'IntermediateImages', _stamp())
            self.save_intermediate_path = top
            self.save_intermediate_vis_path = os.path.join(top, 'Softmax')
            self.save_intermediate_label_path = os.path.join(top, 'Labels')
            for d in (self.save_intermediate_vis_path, self.save_intermediate_label_path):
                os.makedirs(d, exist_ok=True)


for _name, _value in _TRAIN_DEFAULTS.items():
    setattr(CTCParams, _name, _value)
for _name, _value in _INFER_DEFAULTS.items():
    setattr(CTCInferenceParams, _name, _value)
