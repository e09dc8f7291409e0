"""Video dataset and loader (reference: /root/reference/dataloader/vid.py:38-134): file lists of one sequence in the
DAVIS layout (JPEGImages / Annotations / FlowFW / FlowBW / Camera under .../Full-Resolution/<seq>/), forward and
backward pairs, the epoch padded to ~200 iterations, DistributedSampler over ranks."""
import configparser
import glob
import os

import torch
from torch.utils.data import DataLoader

from . import vidbase as base_data


class VidDataset(base_data.BaseDataset):
    """Load video observations including images, flow, and silhouette."""

    def __init__(self, opts, filter_key=None, imglist=None, can_frame=0, dframe=1, init_frame=0):
        super().__init__(opts, filter_key=filter_key)
        self.imglist, self.can_frame, self.dframe = imglist, can_frame, dframe
        seqname = imglist[0].split('/')[-2]
        if opts.sil_path == 'none':
            self.masklist = [i.replace('JPEGImages', 'Annotations').replace('.jpg', '.png') for i in imglist]
        else:
            self.masklist = [('%s/%s/%s' % (opts.sil_path, i.split('/')[-2], i.split('/')[-1])).replace('.jpg', '.png')
                             for i in imglist]
        self.camlist = [i.replace('JPEGImages', 'Camera').replace('.jpg', '.txt') for i in imglist]
        sub = '%s/flo-' % seqname if dframe == 1 else '%s_%02d/flo-' % (seqname, dframe)

        def flows(kind):
            return [i.replace('JPEGImages', kind).replace('.jpg', '.pfm').replace('.png', '.pfm').replace('%s/' % seqname, sub)
                    for i in imglist]
        self.flowfwlist, self.flowbwlist = flows('FlowFW'), flows('FlowBW')

        n = len(imglist) - dframe
        base = list(range(n)) + [i + dframe for i in range(n)]
        direct = [1] * n + [0] * n
        half = len(base) // 2                                                   # frame skipping (vid.py:70-73)
        base = base[:half][init_frame::dframe] + base[half:][init_frame::dframe]
        direct = direct[:half][init_frame::dframe] + direct[half:][init_frame::dframe]
        base = [base[0]] + base + [base[-1]]
        direct = [direct[0]] + direct + [direct[-1]]
        fac = (opts.batch_size * opts.ngpu * 200) // len(direct)                # ~200 iterations per epoch
        self.directlist, self.baselist = direct * fac, base * fac
        self.num_imgs = len(self.directlist)
        print('%d paris of images' % self.num_imgs)


def config_path(opts, root='.'):
    """configs/<dataname>.config under `root` (dataloader/vid.py:98 reads 'configs/%s.config' % opts.dataname)."""
    return os.path.join(root, 'configs', '%s.config' % opts.dataname)


def read_config(dataname, root='.'):
    config = configparser.RawConfigParser()
    path = os.path.join(root, 'configs', '%s.config' % dataname)
    if not config.read(path):
        raise FileNotFoundError(path)
    return {k: (str(config.get('data', k)) if k == 'datapath' else int(config.get('data', k)))
            for k in ('datapath', 'dframe', 'can_frame', 'init_frame', 'end_frame')}


def data_loader(opts, shuffle=True, capdata=None, root='.'):
    cfg = read_config(opts.dataname, root)
    dframe, can_frame, init_frame, end_frame = cfg['dframe'], cfg['can_frame'], cfg['init_frame'], cfg['end_frame']
    datapath = cfg['datapath'] if os.path.isabs(cfg['datapath']) else os.path.join(root, cfg['datapath'])
    imglist = sorted(glob.glob('%s/*' % datapath))
    if not imglist:
        raise FileNotFoundError('no images under %s' % datapath)
    if end_frame > 0:
        imglist = imglist[:end_frame]
    length = (len(imglist) - init_frame) // dframe
    if capdata is not None and capdata < length:                                # window around the canonical frame
        bfac = ffac = capdata // 2
        if can_frame + ffac * dframe > len(imglist):
            ffac = (len(imglist) - can_frame) // dframe
            bfac = capdata - ffac - 1
        if can_frame - bfac * dframe < init_frame:
            bfac = (can_frame - init_frame) // dframe
            ffac = capdata - bfac - 1
        init_frame, end_frame = can_frame - bfac * dframe, can_frame + ffac * dframe
        imglist = imglist[:end_frame + 1]
    print('init:%d, end:%d' % (init_frame, end_frame))
    dataset = VidDataset(opts, imglist=imglist, can_frame=can_frame, dframe=dframe, init_frame=init_frame)
    sampler = torch.utils.data.distributed.DistributedSampler(dataset, num_replicas=opts.ngpu, rank=opts.local_rank,
                                                              shuffle=shuffle)
    # the decoded pairs are cached inside the dataset object, so the loader stays in-process (worker processes would
    # each rebuild the cache and ship every batch through IPC); pinned staging buffers make the H2D copies asynchronous
    loader = DataLoader(dataset, batch_size=opts.batch_size, num_workers=0, drop_last=True,
                        pin_memory=torch.cuda.is_available(), sampler=sampler)
    return loader, length
