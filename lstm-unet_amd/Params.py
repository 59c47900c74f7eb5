"""Run configuration with the field names and defaults of the reference's Params.py
(CTCParams :27-156, CTCInferenceParams :159-195).  Class attributes are the defaults; a dict of
command-line overrides is applied on top (unknown keys warn, Params.py:17-23).

Deviations, on purpose: the default data provider is the synthetic clip stream (the CTC RAM reader is
a SURVEY §8f-2 'next' row) and --root_data_dir is honoured (the reference ignores it, Params.py:104-107).
"""
import os
from datetime import datetime

import DataHandeling
import Networks as Nets

ROOT_DATA_DIR = '~/CellTrackingChallenge/Training/'
ROOT_TEST_DATA_DIR = '~/CellTrackingChallenge/Test/'
ROOT_SAVE_DIR = '~/LSTM-UNet-Outputs/'


class ParamsBase(object):
    aws = False

    def _override_params_(self, params_dict: dict):
        known = set()
        for klass in type(self).__mro__:
            known.update(vars(klass).keys())
        for key, val in params_dict.items():
            if key not in known:
                print('Warning!: Parameter:{} not in defualt parameters'.format(key))
            setattr(self, key, val)


def _lstm_unet_kernels(k_conv, k_lstm):
    widths = (128, 256, 256, 512)
    return {
        'down_conv_kernels': [[(k_conv, w), (k_conv, w)] for w in widths],
        'lstm_kernels': [[(k_lstm, w)] for w in widths],
        'up_conv_kernels': [[(k_conv, 256), (k_conv, 256)], [(k_conv, 128), (k_conv, 128)],
                            [(k_conv, 64), (k_conv, 64)], [(k_conv, 32), (k_conv, 32), (1, 3)]],
    }


class CTCParams(ParamsBase):
    # general
    experiment_name = 'MyRun_SIM'
    gpu_id = 0
    # data
    data_provider_class = DataHandeling.SyntheticSequence2D
    root_data_dir = ROOT_DATA_DIR
    train_sequence_list = [('Fluo-N2DH-SIM+', '01'), ('Fluo-N2DH-SIM+', '02')]
    val_sequence_list = [('Fluo-N2DH-SIM+', '01'), ('Fluo-N2DH-SIM+', '02')]
    crop_size = (128, 128)
    batch_size = 5
    unroll_len = 4
    data_format = 'NCHW'
    train_q_capacity = 200
    val_q_capacity = 200
    num_val_threads = 2
    num_train_threads = 8
    # network: 3x3 encoder/decoder convs, 5x5 ConvLSTM kernels (Params.py:49-69)
    net_model = Nets.ULSTMnet2D
    net_kernel_params = _lstm_unet_kernels(3, 5)
    # training
    class_weights = [0.15, 0.25, 0.6]
    learning_rate = 1e-5
    num_iterations = 1000000
    validation_interval = 1000
    print_to_console_interval = 10
    # save / restore
    load_checkpoint = False
    load_checkpoint_path = ''
    continue_run = False
    save_checkpoint_dir = ROOT_SAVE_DIR
    save_checkpoint_iteration = 5000
    save_checkpoint_every_N_hours = 24
    save_checkpoint_max_to_keep = 5
    # logging
    tb_sub_folder = 'LSTMUNet'
    write_to_tb_interval = 500
    save_log_dir = ROOT_SAVE_DIR
    # debugging
    dry_run = False
    profile = False
    # MI355X additions
    sync_bn = False

    def __init__(self, params_dict):
        self._override_params_(params_dict)
        root = os.path.expanduser(self.root_data_dir)
        self.train_data_base_folders = [(os.path.join(root, ds[0]), ds[1]) for ds in self.train_sequence_list]
        self.val_data_base_folders = [(os.path.join(root, ds[0]), ds[1]) for ds in self.val_sequence_list]
        rank = int(os.environ.get('RANK', '0'))
        common = dict(image_crop_size=self.crop_size, unroll_len=self.unroll_len, deal_with_end=0,
                      batch_size=self.batch_size, data_format=self.data_format, randomize=True, return_dist=False)
        self.train_data_provider = self.data_provider_class(sequence_folder_list=self.train_data_base_folders,
                                                            queue_capacity=self.train_q_capacity,
                                                            num_threads=self.num_train_threads, **common)
        self.val_data_provider = self.data_provider_class(sequence_folder_list=self.val_data_base_folders,
                                                          queue_capacity=self.train_q_capacity,
                                                          num_threads=self.num_val_threads, **common)
        now_string = datetime.now().strftime('%Y-%m-%d_%H%M%S')
        if self.load_checkpoint and self.continue_run:
            path = self.load_checkpoint_path
            base = path if os.path.isdir(path) else os.path.dirname(os.path.dirname(path))
            if base.rstrip('/').endswith('tf-ckpt'):
                base = os.path.dirname(base.rstrip('/'))
            self.experiment_log_dir = self.experiment_save_dir = base
        else:
            self.experiment_log_dir = os.path.join(os.path.expanduser(self.save_log_dir), self.tb_sub_folder,
                                                   self.experiment_name, now_string)
            self.experiment_save_dir = os.path.join(os.path.expanduser(self.save_checkpoint_dir), self.tb_sub_folder,
                                                    self.experiment_name, now_string)
        if not self.dry_run and rank == 0:
            for d in (self.experiment_log_dir, self.experiment_save_dir, os.path.join(self.experiment_log_dir, 'train'),
                      os.path.join(self.experiment_log_dir, 'val')):
                os.makedirs(d, exist_ok=True)
        self.channel_axis = 1 if self.data_format == 'NCHW' else 3


class CTCInferenceParams(ParamsBase):
    gpu_id = 0
    model_path = './Models/LSTMUNet2D/PhC-C2DL-PSC/'
    output_path = './tmp/output/PhC-C2DL-PSC/01'
    sequence_path = os.path.join(ROOT_TEST_DATA_DIR, 'PhC-C2DL-PSC/01/')
    filename_format = 't*.tif'
    data_reader = DataHandeling.CTCInferenceReader
    data_format = 'NCHW'
    FOV = 0
    min_cell_size = 10
    max_cell_size = 100
    edge_dist = 2
    pre_sequence_frames = 4
    dry_run = False
    save_intermediate = True
    save_intermediate_path = output_path

    def __init__(self, params_dict: dict = None):
        if params_dict is not None:
            self._override_params_(params_dict)
        self.channel_axis = 1 if self.data_format == 'NCHW' else 3
        if not self.dry_run:
            os.makedirs(self.output_path, exist_ok=True)
            if self.save_intermediate:
                now_string = datetime.now().strftime('%Y-%m-%d_%H%M%S')
                self.save_intermediate_path = os.path.join(self.save_intermediate_path, 'IntermediateImages', now_string)
                self.save_intermediate_vis_path = os.path.join(self.save_intermediate_path, 'Softmax')
                self.save_intermediate_label_path = os.path.join(self.save_intermediate_path, 'Labels')
                for d in (self.save_intermediate_path, self.save_intermediate_vis_path, self.save_intermediate_label_path):
                    os.makedirs(d, exist_ok=True)
