"""Training driver with the command line of the reference's train2D.py (flags :288-352) and its
step contract (train_step/val_step :87-117, loop :145-220):

  one step = forward(training=True) -> WeightedCELoss on the logits -> gradients of every trainable
  tensor -> Adam(lr, .9, .999, 1e-7) -> step counter += 1; the recurrent-state keep-mask of THAT
  batch is applied after the step; validation runs with training=False on its own swapped-in
  recurrent state; both steps return (softmax, logits channels-last, loss).

Multi-GPU: launched as one process per GPU (torchrun); batch slots shard across ranks and gradient
buckets are all-reduced by RCCL while backward runs (lu_native.dp).  The reference is single-device
(--net_gpus is parsed and unused there, train2D.py:305); the flag is kept and ignored here too.
"""
import argparse
import os
import pickle
import sys
import time

import numpy as np
import torch

import Networks as Nets
import losses
from lu_native import ops
from lu_native.dp import DataParallel
from lu_native.engine import Adam
from utils import log_print, select_gpu


class AgreedFailure(Exception):
    """A failure every data-parallel rank raises at the same loop position (train(): agree_on_failure).  `handled`: the error
    that started it is of a type the reference saves and closes on (ValueError from a reader, the spot-instance notice,
    train2D.py:222-227) -- every rank checkpoints and returns normally; otherwise every rank checkpoints and RE-RAISES, so a
    launcher sees one outcome for one failure.  The rank that failed carries its own exception as __cause__."""

    def __init__(self, message, handled):
        super(AgreedFailure, self).__init__(message)
        self.handled = bool(handled)


class AWSError(Exception):
    pass


class Trainer(object):
    """Owns model + optimiser + DP plumbing; exposes train_step / val_step."""

    def __init__(self, net_model, net_kernel_params, data_format='NCHW', class_weights=(0.15, 0.25, 0.6),
                 learning_rate=1e-5, dp=None, sync_bn=False, seed=0, precision='fp32', resize='tf2.0'):
        self.dp = dp if dp is not None else DataParallel()
        self.data_format = data_format
        self.model = net_model(net_kernel_params, data_format, False, seed=seed, dp=self.dp, sync_bn=sync_bn,
                               precision=precision, resize=resize)
        self.engine = self.model.engine
        self.optimizer = Adam(self.engine, lr=learning_rate)
        self.class_weights = list(class_weights)
        self._cw = None
        self._nchw = data_format[1] == 'C'
        self.step = 0
        self.engine.on_bucket_ready = self.dp.bucket_ready

    def _prep(self, image, label):
        dev = Nets._device()
        x_tb, T, B = Nets._to_internal(image, self._nchw, dev)
        g_tb, _, _ = Nets._to_internal(label, self._nchw, dev)
        if self._cw is None:
            self._cw = torch.tensor(self.class_weights, dtype=torch.float32, device=dev)
        return x_tb, g_tb.view(-1), T, B

    def _outputs(self, logits_tb, sm_tb, T, B):
        softmax = Nets._from_internal(sm_tb.view(logits_tb.shape), T, B, self._nchw)
        predictions = Nets._from_internal(logits_tb, T, B, False)      # channels-last, as train2D.py:98-100
        return softmax, predictions

    def train_step(self, image, label, want_outputs=True):
        x_tb, g, T, B = self._prep(image, label)
        e = self.engine
        logits = e.forward(x_tb, T, B, True)
        if self.dp.flat is None:
            self.dp.attach(e.flat_grads)
        lg = logits.view(-1, logits.shape[-1])
        sums, sm = ops.wce_forward(lg, g, self._cw, want_outputs)
        self.dp.all_reduce_(sums)            # global valid-pixel normalisation (losses.py:26)
        dl = ops.wce_backward(lg, g, self._cw, sums, 1.0)
        e.backward(dl.view(logits.shape))
        self.dp.finish()
        self.optimizer.apply_gradients(1.0)  # gradients were SUMMED over ranks of a globally normalised loss
        self.step += 1
        loss = ops.wce_loss(sums)
        if not want_outputs:
            return None, None, loss
        softmax, predictions = self._outputs(logits, sm, T, B)
        return softmax, predictions, loss

    def val_step(self, image, label):
        x_tb, g, T, B = self._prep(image, label)
        logits = self.engine.forward(x_tb, T, B, False)
        lg = logits.view(-1, logits.shape[-1])
        sums, sm = ops.wce_forward(lg, g, self._cw, True)
        self.dp.all_reduce_(sums)
        softmax, predictions = self._outputs(logits, sm, T, B)
        return softmax, predictions, ops.wce_loss(sums)

    # -- checkpoints (own format) --
    def state_dict(self):
        e = self.engine
        return {'step': self.step, 'params': e.flat_params.cpu(), 'bn': {k: v.cpu() for k, v in e.S.items()},
                'adam_m': None if self.optimizer.m is None else self.optimizer.m.cpu(),
                'adam_v': None if self.optimizer.v is None else self.optimizer.v.cpu(),
                'adam_iterations': self.optimizer.iterations,
                'states': self.states_for_checkpoint()}

    def states_for_checkpoint(self):
        """The recurrent (h, c) of every ConvLSTM as host tensors -- all a non-zero rank writes under data parallelism."""
        return [[[None, None] if st[0] is None else [torch.from_numpy(st[0]), torch.from_numpy(st[1])]
                 for st in blk] for blk in self.model.get_states()]

    def load_state_dict(self, sd, in_channels=1):
        e = self.engine
        e.build(in_channels, Nets._device())
        with torch.no_grad():      # flat_params may be an autograd leaf (Networks.parameters())
            e.flat_params.copy_(sd['params'])
        for k, v in sd['bn'].items():
            e.S[k].copy_(v)
        if sd['adam_m'] is not None:
            self.optimizer.m = sd['adam_m'].to(e.device)
            self.optimizer.v = sd['adam_v'].to(e.device)
        self.optimizer.iterations = sd['adam_iterations']
        self.step = sd['step']
        if sd.get('states') is not None:
            self.model.set_states(sd['states'])


class _RunningMean(object):
    """k.metrics.Mean: never reset in the reference (train2D.py:52-58) -> running mean over the run."""

    def __init__(self):
        self.total, self.count = 0.0, 0

    def __call__(self, v):
        v = float(v)
        if not np.isnan(v):
            self.total += v
            self.count += 1

    def result(self):
        return self.total / max(self.count, 1)


def train(params):
    select_gpu(getattr(params, 'gpu_id', None))
    dp = DataParallel()
    is_main = dp.rank == 0
    trainer = Trainer(params.net_model, params.net_kernel_params, params.data_format, params.class_weights,
                      params.learning_rate, dp=dp, sync_bn=getattr(params, 'sync_bn', False),
                      precision=getattr(params, 'precision', 'fp32'), resize=getattr(params, 'resize', 'tf2.0'))
    model = trainer.model
    if dp.world_size > 1:                    # one run directory for the job: rank 0's (the time stamp is per process)
        dirs = [params.experiment_log_dir, params.experiment_save_dir]
        torch.distributed.broadcast_object_list(dirs, src=0)
        for own, shared in zip((params.experiment_log_dir, params.experiment_save_dir), dirs):
            # Params.__init__ has already created this rank's own time-stamped directories: drop the unused (empty) ones
            if own != shared and os.path.isdir(own) and not os.listdir(own):
                try:
                    os.rmdir(own)
                except OSError:
                    pass
        params.experiment_log_dir, params.experiment_save_dir = dirs
    train_data_provider, val_data_provider = params.train_data_provider, params.val_data_provider
    train_data_provider.start_queues(None)
    val_data_provider.start_queues(None)
    seg_measure = losses.seg_measure(params.channel_axis + 1, three_d=False)
    train_loss, train_seg, train_acc = _RunningMean(), _RunningMean(), _RunningMean()
    val_loss, val_seg, val_acc = _RunningMean(), _RunningMean(), _RunningMean()
    ckpt_dir = os.path.join(params.experiment_save_dir, 'tf_ckpts')
    saved = []

    def sync_bn_stats():
        """Rank-local BatchNorm (the default under DP) lets every rank drift its own moving statistics; what is saved
        (and what every rank continues from) is their mean over the ranks (SURVEY §8e semantics decision 3)."""
        if dp.world_size > 1 and not trainer.engine.sync_bn and trainer.engine.plan is not None:
            for t in trainer.engine.S.values():
                dp.all_reduce_(t)
                t.mul_(1.0 / dp.world_size)

    saved_states = []
    bn_synced_at = [None]                    # step whose BatchNorm statistics are already the mean over the ranks

    def save_ckpt(collective=True):
        """collective=True: a regular save point that EVERY rank reaches at the same step (BN statistics are averaged over
        the ranks first -- an all-reduce).  collective=False: the error path, which may run on ONE rank only (SIGINT to one
        pid, a spot-instance notice on one node ...): no collective may be issued there -- the peers sit in other
        all-reduces -- so the rank writes what it has, rank-local."""
        if params.dry_run:
            return None
        if collective:
            sync_bn_stats()
            bn_synced_at[0] = trainer.step
        if dp.world_size > 1:                # recurrent states are rank-local clip streams: one small file per rank
            os.makedirs(ckpt_dir, exist_ok=True)
            own = os.path.join(ckpt_dir, 'states-%d.rank%d.pt' % (trainer.step, dp.rank))
            torch.save(trainer.states_for_checkpoint(), own)
            if own not in saved_states:
                saved_states.append(own)
            while len(saved_states) > params.save_checkpoint_max_to_keep:      # same rotation as ckpt-*.pt
                old = saved_states.pop(0)
                if os.path.exists(old):
                    os.remove(old)
        if not is_main:
            return None
        os.makedirs(ckpt_dir, exist_ok=True)
        path = os.path.join(ckpt_dir, 'ckpt-%d.pt' % trainer.step)
        torch.save(trainer.state_dict(), path)
        if path not in saved:                # (the error path may re-write the checkpoint of the step just saved)
            saved.append(path)
        while len(saved) > params.save_checkpoint_max_to_keep:
            old = saved.pop(0)
            if os.path.exists(old):
                os.remove(old)
        return path

    if params.load_checkpoint:
        path = params.load_checkpoint_path
        if os.path.isdir(path):
            cands = sorted((f for f in os.listdir(path) if f.startswith('ckpt-')),
                           key=lambda f: int(f.split('-')[1].split('.')[0]))
            path = os.path.join(path, cands[-1]) if cands else ''
        if path:
            try:
                sd = torch.load(path, map_location='cpu')
                if dp.world_size > 1:            # this rank's own clip-stream states, or none (zeros) if it has no file
                    own = os.path.join(os.path.dirname(path), 'states-%d.rank%d.pt' % (sd['step'], dp.rank))
                    if os.path.exists(own):
                        sd['states'] = torch.load(own, map_location='cpu')
                    else:
                        log_print('rank {}: no recurrent-state file {} -- its clip streams restart from zero state'.format(
                            dp.rank, own))
                        sd['states'] = None
                trainer.load_state_dict(sd)
                log_print('Restored from {}'.format(path))
            except FileNotFoundError:
                raise ValueError('Could not load checkpoint: {}'.format(path))
        else:
            log_print('Initializing from scratch.')
    else:
        log_print('Initializing from scratch.')

    def accuracy(label, predictions):
        lab = torch.as_tensor(label).squeeze(params.channel_axis + 1)
        pred = predictions.argmax(-1).cpu()
        return float((pred == lab.long()).float().mean())

    # TensorBoard event files (train2D.py:119-137,163-176,205-211): scalars Loss / SEG, images Image / GT / Output of the
    # window's last frame (first batch slot), every write_to_tb_interval steps -- written by rank 0 only
    writers = None
    if not params.dry_run and is_main:
        import tb_events
        writers = {'train': tb_events.SummaryWriter(os.path.join(params.experiment_log_dir, 'train')),
                   'val': tb_events.SummaryWriter(os.path.join(params.experiment_log_dir, 'val'))}

    def tboard(which, step, loss_mean, seg_mean, image, label, softmax_):
        w = writers[which]
        w.scalar('Loss', loss_mean.result(), step)
        w.scalar('SEG', seg_mean.result(), step)
        nchw = params.channel_axis == 1
        img = np.asarray(image)[0, -1]
        img = img[0] if nchw else img[..., 0]
        img = img - img.min()
        w.image('Image', img / max(float(img.max()), 1e-12), step)
        lab = np.asarray(label)[0, -1]
        lab = (lab[0] if nchw else lab[..., 0]).astype(np.int64)
        w.image('GT', np.eye(3, dtype=np.float32)[np.clip(lab, 0, 2)] * (lab >= 0)[..., None], step)
        sm_ = softmax_[0, -1].detach().cpu().numpy()
        w.image('Output', np.transpose(sm_, (1, 2, 0)) if sm_.shape[0] == 3 else sm_, step)
        w.flush()

    template = '{}: Step {}, Loss: {}, Accuracy: {}'
    val_states = model.get_states()
    # True while every rank is known to be at the same point of the loop: normal completion, or a failure that the per-step
    # flag all-reduce has spread to all ranks.  Only then may the shutdown path issue collectives.
    in_step_with_peers = False

    def agree_on_failure(local_err):
        """One two-word all-reduce: a rank that fails (data error, spot-instance notice) takes the others with it at the same
        loop position instead of leaving them in a gradient all-reduce.  The words count the failing ranks by KIND -- an error
        type the reference handles (save, close, return) or any other (save, re-raise) -- so that all ranks end the same way."""
        if local_err is None and dp.world_size == 1:
            return
        mine_handled = isinstance(local_err, (ValueError, AWSError))
        n_handled, n_other = (1.0 if (local_err is not None and mine_handled) else 0.0,
                              1.0 if (local_err is not None and not mine_handled) else 0.0)
        if dp.world_size > 1:
            flag = torch.tensor([n_handled, n_other], device=Nets._device())
            dp.all_reduce_(flag)
            n_handled, n_other = (float(v) for v in flag.cpu())
        if n_handled + n_other == 0:
            return
        what = 'this rank: %s: %s' % (type(local_err).__name__, local_err) if local_err is not None else \
            'another data-parallel rank reported an error'
        raise AgreedFailure(what, handled=(n_other == 0)) from local_err      # every rank raises here, at the same loop position

    try:
        for _ in range(trainer.step, params.num_iterations + 1):
            local_err = None
            if params.aws:
                import requests
                r = requests.get('http://169.254.169.254/latest/meta-data/spot/instance-action')
                if not r.status_code == 404:
                    local_err = AWSError('Quitting Spot Instance Gracefully')
            if local_err is None:
                try:
                    image_sequence, seg_sequence, _, is_last_batch = train_data_provider.get_batch()
                except Exception as exc:      # ValueError, but also the reader's RuntimeError (workers stopped / died) and a
                    local_err = exc           # producer's own OSError / IndexError: every type must reach the flag all-reduce
            agree_on_failure(local_err)
            profiling = bool(params.profile) and is_main and not params.dry_run and \
                (trainer.step + 1) % params.write_to_tb_interval == 0
            if profiling:                        # --profile (train2D.py:152-160): this step under per-kernel HIP events
                ops.EVENT_LOG = []
            softmax, predictions, loss_value = trainer.train_step(image_sequence, seg_sequence)
            if profiling:
                import json
                from lu_native.profile import summarize_events
                torch.cuda.synchronize()
                ev, ops.EVENT_LOG = ops.EVENT_LOG, None
                mfma_rows, hbm_rows = summarize_events(ev)
                prof_dir = os.path.join(params.experiment_log_dir, 'profile')
                os.makedirs(prof_dir, exist_ok=True)
                with open(os.path.join(prof_dir, 'step_%d.json' % trainer.step), 'w') as fh:
                    json.dump({'step': trainer.step, 'mfma_kernels': mfma_rows, 'hbm_kernels': hbm_rows,
                               'counters': 'run the same command under `rocprofv3 --kernel-trace --stats` / `--pmc '
                                           'FETCH_SIZE` / `--pmc WRITE_SIZE` / `--pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE` '
                                           'for HBM bytes and MFMA utilisation (tools/gpu/pmc_traffic.sh)'}, fh, indent=1)
                log_print('Profiled step {}: {}'.format(trainer.step, ', '.join(
                    '%s %.0f %s' % (r['kernel'].split(' ')[0], r['achieved'], r['unit']) for r in (mfma_rows + hbm_rows)[:4])))
            model.reset_states_per_batch(is_last_batch)
            train_loss(loss_value)
            pred_pub = predictions.permute(0, 1, 4, 2, 3) if params.channel_axis == 1 else predictions
            train_seg(seg_measure(seg_sequence, pred_pub))
            train_acc(accuracy(seg_sequence, predictions))
            step = trainer.step
            if step % params.save_checkpoint_iteration == 0 or step == params.num_iterations:
                p = save_ckpt()
                if p:
                    log_print('Saved checkpoint for step {}: {}'.format(step, p))
            if writers is not None and not step % params.write_to_tb_interval:
                tboard('train', step, train_loss, train_seg, image_sequence, seg_sequence, softmax)
                log_print('Printed Training Step: {} to Tensorboard'.format(step))
            if not step % params.print_to_console_interval and is_main:
                log_print(template.format('Training', step, train_loss.result(), train_acc.result() * 100))
            if not step % params.validation_interval:
                train_states = model.get_states()
                model.set_states(val_states)
                local_err = None
                try:
                    v_img, v_seg, _, v_last = val_data_provider.get_batch()
                except Exception as exc:
                    local_err = exc
                agree_on_failure(local_err)
                v_sm, v_pred, v_loss = trainer.val_step(v_img, v_seg)
                model.reset_states_per_batch(v_last)
                val_loss(v_loss)
                v_pub = v_pred.permute(0, 1, 4, 2, 3) if params.channel_axis == 1 else v_pred
                val_seg(seg_measure(v_seg, v_pub))
                val_acc(accuracy(v_seg, v_pred))
                if is_main:
                    log_print(template.format('Validation', step, val_loss.result(), val_acc.result() * 100))
                if writers is not None and not step % params.write_to_tb_interval:
                    tboard('val', step, val_loss, val_seg, v_img, v_seg, v_sm)
                val_states = model.get_states()
                model.set_states(train_states)
        in_step_with_peers = True                # the loop ran to its end on every rank
    except BaseException as err:
        # the reference saves and closes on these three (train2D.py:222-227); a reader failure of any other type that
        # agree_on_failure has spread to all ranks is checkpointed the same way and then re-raised
        agreed = isinstance(err, AgreedFailure)
        handled = err.handled if agreed else isinstance(err, (KeyboardInterrupt, ValueError, AWSError))
        if not handled and not agreed:
            raise
        # agree_on_failure raised on every rank at once; anything else (SIGINT to one pid, an error inside a step) may be
        # this rank's alone -- no collectives from here on in that case
        in_step_with_peers = dp.world_size == 1 or agreed
        # a single process has nobody to agree with: the reader's own exception (type and text) is what the caller and the log see,
        # as before AgreedFailure existed; with peers, AgreedFailure carries it as __cause__ on the rank that failed
        own = err.__cause__ if (agreed and dp.world_size == 1 and err.__cause__ is not None) else None
        if not params.dry_run:
            log_print('Saving Model Before closing due to error: {}'.format(str(own if own is not None else err)))
            # all ranks in step: the resumable checkpoint gets the rank-averaged BatchNorm statistics like a regular one
            save_ckpt(collective=in_step_with_peers and dp.world_size > 1)
        if not handled:
            if own is not None:
                raise own from None
            raise
    finally:
        if not params.dry_run and trainer.engine.plan is not None and in_step_with_peers and bn_synced_at[0] != trainer.step:
            sync_bn_stats()
        if not params.dry_run and is_main and trainer.engine.plan is not None:
            model_fname = os.path.join(params.experiment_save_dir, 'model.ckpt')
            # as the reference does (train2D.py:235): a TensorFlow tensor bundle model.ckpt.index + .data-00000-of-00001, which
            # the reference's own Inference2D.py (and ours) loads; params.save_format = 'pt' writes a torch blob instead
            model.save_weights(model_fname, save_format=getattr(params, 'save_format', 'tf'))
            with open(os.path.join(params.experiment_save_dir, 'model_params.pickle'), 'wb') as fobj:
                # 'name' / 'params' are the reference's contract (train2D.py:236-239); 'resize' / 'precision' record what the
                # weights were trained under -- the bilinear convention is part of the function they encode
                pickle.dump({'name': model.__class__.__name__, 'params': (params.net_kernel_params,),
                             'resize': trainer.engine.resize, 'precision': trainer.engine.precision}, fobj,
                            protocol=pickle.HIGHEST_PROTOCOL)
            log_print('Saved Model to file: {}'.format(model_fname))
        elif params.dry_run:
            log_print('WARNING: dry_run flag is ON! Not Saving Model')
        if writers is not None:
            for w_ in writers.values():
                w_.close()
        log_print('Done')
    return trainer


# ------------------------------------------------------------------------------------------------
# command line: same 32 flags / dests as the reference (train2D.py:288-352)
# ------------------------------------------------------------------------------------------------
class _AddNets(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        setattr(namespace, self.dest, [getattr(Nets, v) for v in values])


class _AddReader(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        import DataHandeling
        setattr(namespace, self.dest, getattr(DataHandeling, values))


class _AddDatasets(argparse.Action):
    def __call__(self, parser, namespace, values, option_string=None):
        if len(values) % 2:
            raise ValueError('dataset values should be of length 2*N where N is the number of datasets')
        setattr(namespace, self.dest, [(values[i], values[i + 1]) for i in range(0, len(values), 2)])


_FLAG = dict  # readability
FLAGS = [
    (('-n', '--experiment_name'), _FLAG(dest='experiment_name', type=str, help='Name of experiment')),
    (('--gpu_id',), _FLAG(dest='gpu_id', type=str, help="Visible GPUs: example, '0,2,3'")),
    (('--dry_run',), _FLAG(dest='dry_run', action='store_const', const=True, help='Do not write any outputs')),
    (('--profile',), _FLAG(dest='profile', type=bool,
                           help='Write per-kernel timing / roofline tables of one step every write_to_tb_interval steps to <log_dir>/profile')),
    (('--root_data_dir',), _FLAG(dest='root_data_dir', type=str, help='Root folder containing training data')),
    (('--data_provider_class',), _FLAG(dest='data_provider_class', type=str, action=_AddReader,
                                       help='Type of data provider')),
    (('--dataset',), _FLAG(dest='train_sequence_list', type=str, action=_AddDatasets, nargs='+',
                           help='Datasets to run. string of pairs: DatasetName, SequenceNumber')),
    (('--val_dataset',), _FLAG(dest='val_sequence_list', type=str, action=_AddDatasets, nargs='+',
                               help='Datasets to run. string of pairs DatasetName, SequenceNumber')),
    (('--net_gpus',), _FLAG(dest='net_gpus', type=int, nargs='+', help='gpus for each net (unused, as upstream)')),
    (('--net_types',), _FLAG(dest='net_types', type=int, nargs='+', action=_AddNets, help='Type of nets')),
    (('--crop_size',), _FLAG(dest='crop_size', type=int, nargs=2, help='crop size for y and x dimensions')),
    (('--train_q_capacity',), _FLAG(dest='train_q_capacity', type=int, help='Capacity of training queue')),
    (('--val_q_capacity',), _FLAG(dest='val_q_capacity', type=int, help='Capacity of validation queue')),
    (('--num_train_threads',), _FLAG(dest='num_train_threads', type=int, help='Number of train data threads')),
    (('--num_val_threads',), _FLAG(dest='num_val_threads', type=int, help='Number of validation data threads')),
    (('--data_format',), _FLAG(dest='data_format', type=str, choices=['NCHW', 'NWHC', 'NHWC'],
                               help="Data format NCHW or NHWC ('NWHC' accepted: upstream typo)")),
    (('--batch_size',), _FLAG(dest='batch_size', type=int, help='Batch size')),
    (('--unroll_len',), _FLAG(dest='unroll_len', type=int, help='LSTM unroll length')),
    (('--num_iterations',), _FLAG(dest='num_iterations', type=int, help='Maximum number of training iterations')),
    (('--validation_interval',), _FLAG(dest='validation_interval', type=int,
                                       help='Number of iterations between validation iteration')),
    (('--load_checkpoint',), _FLAG(dest='load_checkpoint', action='store_const', const=True,
                                   help='Load from checkpoint')),
    (('--load_checkpoint_path',), _FLAG(dest='load_checkpoint_path', type=str,
                                        help='path to checkpoint, used only with --load_checkpoint')),
    (('--continue_run',), _FLAG(dest='continue_run', action='store_const', const=True,
                                help='Continue run in existing directory')),
    (('--learning_rate',), _FLAG(dest='learning_rate', type=float, help='Learning rate')),
    (('--class_weights',), _FLAG(dest='class_weights', type=float, nargs=3,
                                 help='class weights for background, foreground and edge classes')),
    (('--save_checkpoint_dir',), _FLAG(dest='save_checkpoint_dir', type=str, help='root directory to save checkpoints')),
    (('--save_log_dir',), _FLAG(dest='save_log_dir', type=str, help='root directory to save logs')),
    (('--tb_sub_folder',), _FLAG(dest='tb_sub_folder', type=str, help='sub-folder to save outputs')),
    (('--save_checkpoint_iteration',), _FLAG(dest='save_checkpoint_iteration', type=int,
                                             help='number of iterations between save checkpoint')),
    (('--save_checkpoint_max_to_keep',), _FLAG(dest='save_checkpoint_max_to_keep', type=int,
                                               help='max recent checkpoints to keep')),
    (('--save_checkpoint_every_N_hours',), _FLAG(dest='save_checkpoint_every_N_hours', type=int,
                                                 help='keep checkpoint every N hours')),
    (('--write_to_tb_interval',), _FLAG(dest='write_to_tb_interval', type=int, help='Interval between log writes')),
]


def build_arg_parser():
    parser = argparse.ArgumentParser(description='Run Train LSTMUnet Segmentation (MI355X-native)')
    for names, kw in FLAGS:
        parser.add_argument(*names, **kw)
    parser.add_argument('--sync_bn', dest='sync_bn', action='store_const', const=True,
                        help='[MI355X] pool BatchNorm statistics over all DP ranks')
    parser.add_argument('--data_provider', dest='data_provider', choices=['synthetic', 'ctc'],
                        help='[MI355X] synthetic clips (default) or the Cell-Tracking-Challenge RAM reader over '
                             '--root_data_dir / --train_sequence_list (needs the metadata_<seq>.pickle files)')
    parser.add_argument('--precision', dest='precision', choices=['fp32', 'bf16', 'bf16x3'],
                        help='[MI355X] fp32 (default) or bf16-MFMA operands for the wide stride-1 convolutions')
    parser.add_argument('--resize', dest='resize', choices=['tf2.0', 'half_pixel'],
                        help="[MI355X] bilinear convention of the up blocks: 'tf2.0' (default; TensorFlow 2.0 / 2.1, the release the "
                             "reference pins) or 'half_pixel' (later releases); recorded in model_params.pickle")
    return parser


if __name__ == '__main__':
    import Params
    args = build_arg_parser().parse_args()
    args_dict = {key: val for key, val in vars(args).items() if val is not None}
    print(args_dict)
    train(Params.CTCParams(args_dict))
