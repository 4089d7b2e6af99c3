"""TEST INFRASTRUCTURE ONLY (never imported by papc_amd/): a literal restatement of the reference's ShapeNet-part loaders
(/root/reference/PAPC/datasets/pnloader.py:7-106), statement by statement -- per-sample lists, per-sample transpose and dtype conversion, the
global ``random`` module -- with the file access replaced by ``opener(path)`` (h5py is not in this image; a mapping with ['data'] / ['label'] /
['pid'] arrays stands for an open h5py.File).  PARITY UNPINNED like the rest of the oracle: the reference holds no loader tests or fixtures.
"""
import os
import random

import numpy as np

train_list = ['ply_data_train0.h5', 'ply_data_train1.h5', 'ply_data_train2.h5', 'ply_data_train3.h5', 'ply_data_train4.h5', 'ply_data_train5.h5']   # datalist.py:1
test_list = ['ply_data_test0.h5', 'ply_data_test1.h5']      # datalist.py:2
val_list = ['ply_data_val0.h5']                              # datalist.py:3


def PNClasDataLoader(opener, max_point=1024, batchsize=64, path='./data/', mode='train'):
    datas = []
    labels = []
    files = train_list if mode == 'train' else (test_list if mode == 'test' else val_list)    # pnloader.py:12-31 (three identical branches)
    for file_list in files:
        f = opener(os.path.join(path, file_list))                                              # :14
        datas.extend(f['data'][:, :max_point, :])                                              # :15
        labels.extend(f['label'])                                                              # :16
    datas = np.array(datas)                                                                    # :33
    labels = np.array(labels)                                                                  # :34
    index_list = list(range(len(datas)))                                                       # :37

    def PNClasDataGenerator():
        if mode == 'train':
            random.shuffle(index_list)                                                         # :40-41
        datas_list = []
        labels_list = []
        for i in index_list:                                                                   # :44
            datas_list.append(datas[i].T.astype('float32'))                                    # :45
            labels_list.append(labels[i].astype('int64'))                                      # :46
            if len(datas_list) == batchsize:                                                   # :47
                yield np.array(datas_list), np.array(labels_list)                              # :48
                datas_list = []
                labels_list = []
        if len(datas_list) > 0:                                                                # :51
            yield np.array(datas_list), np.array(labels_list)

    return PNClasDataGenerator


def PNSegDataLoader(opener, max_point=1024, batchsize=64, path='./data/', mode='train'):
    datas = []
    labels = []
    targets = []
    files = train_list if mode == 'train' else (test_list if mode == 'test' else val_list)    # :60-81
    for file_list in files:
        f = opener(os.path.join(path, file_list))
        datas.extend(f['data'][:, :max_point, :])                                              # :63
        labels.extend(f['label'])                                                              # :64
        targets.extend(f['pid'][:, :max_point])                                                # :65
    datas = np.array(datas)
    labels = np.array(labels)
    targets = np.array(targets)
    index_list = list(range(len(datas)))                                                       # :88

    def PNSegDataGenerator():
        if mode == 'train':
            random.shuffle(index_list)                                                         # :91-92
        datas_list = []
        labels_list = []
        targets_list = []
        for i in index_list:
            target = np.reshape(targets[i], [max_point, -1]).astype('int64')                   # :97
            datas_list.append(datas[i].T.astype('float32'))                                    # :98
            labels_list.append(labels[i].astype('int64'))                                      # :99
            targets_list.append(target)                                                        # :100
            if len(datas_list) == batchsize:
                yield [np.array(datas_list), np.array(labels_list)], np.array(targets_list)    # :102
                datas_list = []
                labels_list = []
                targets_list = []
        if len(datas_list) > 0:
            yield [np.array(datas_list), np.array(labels_list)], np.array(targets_list)        # :107

    return PNSegDataGenerator
