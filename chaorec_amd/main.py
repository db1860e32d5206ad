"""Entry point: `python -m chaorec_amd.main --Model LightGCN --data_path sports` (reference main.py:73-442).

Same flow: parse flags, seed, load data, build the sampler, cartesian product over the model's YAML grid,
construct the model, Adam, train_and_evaluate, keep the best combination by Recall@20.  Out of scope and
dropped: the 51 other constructors and the dense U x I DiffusionData the reference builds for every model
(main.py:202; 1.76 GB for sports)."""
import logging
import os
from itertools import product

import torch
from torch.utils.data import DataLoader

from . import dataload
from .Model import (BM3, BPRMF, DCCF, DDRec, DHCF, FKAN_GCF, FREEDOM, GRCN, GUME, HCCF, LATTICE, LGMRec, LightGCL, LayerGCN, LightGCN, LightGT, MCLN, MENTOR, MGAT, MGCL, MGCN, MICRO, MMGCN, MMSSL, NCL, NGCF, MMGCL, POWERec, SelfCF, SGL, SimGCL, SLMRec,
                    SMORE, VBPR, VGCL, XSimGCL)
from .arg_parser import load_yaml_config, parse_args
from .train_and_evaluate import train_and_evaluate
from .optim import FusedAdam
from .utils import get_local_time, gpu, setup_seed


def setup_logging(args):
    log_dir = "log"
    os.makedirs(log_dir, exist_ok=True)
    log_filename = os.path.join(log_dir, f"{args.Model}_{args.data_path}").replace("\\", "/") + ".log"
    fmt = logging.Formatter('%(asctime)s %(levelname)s %(message)s', '%a %d %b %Y %H:%M:%S')
    logger = logging.getLogger()
    logger.setLevel(logging.INFO)
    for h in (logging.StreamHandler(), logging.FileHandler(log_filename, mode='w')):
        h.setLevel(logging.INFO)
        h.setFormatter(fmt)
        logger.addHandler(h)


def build_model(args, num_user, num_item, train_data, user_item_dict, v_feat, t_feat, device):
    """The rows of the reference's constructor table on the hot path (main.py:261-263, :267-270, :287-289, :316-317, :323-324)."""
    dim_E, aggr_mode = args.dim_E, args.aggr_mode
    table = {
        'MMGCN': lambda: MMGCN(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                               aggr_mode, 'False', True, device),
        'LightGCN': lambda: LightGCN(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight,
                                     args.n_layers, aggr_mode, device),
        'NGCF': lambda: NGCF(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.dropout,
                             args.n_layers, aggr_mode, device),
        'FREEDOM': lambda: FREEDOM(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E,
                                   args.feature_embed, args.reg_weight, args.dropout, args.n_layers, args.mm_layers,
                                   args.ii_topk, args.lambda_coeff, device),
        'LayerGCN': lambda: LayerGCN(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight,
                                     args.n_layers, args.dropout, device),
        'MGCN': lambda: MGCN(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                             args.n_layers, aggr_mode, args.ssl_temp, args.ssl_alpha, device),
        'BPR': lambda: BPRMF(num_user, num_item, user_item_dict, dim_E, args.reg_weight, device),
        'VBPR': lambda: VBPR(num_user, num_item, user_item_dict, v_feat, dim_E, args.feature_embed, args.reg_weight,
                             device),
        # three more members of the torch.sparse.mm family, through the adapter alone (main.py:305-306, :335-336, :344-345)
        'NCL': lambda: NCL(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers, aggr_mode,
                           args.ssl_temp, args.ssl_alpha, device),
        'SimGCL': lambda: SimGCL(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers,
                                 args.ssl_temp, args.ssl_alpha, device),
        'XSimGCL': lambda: XSimGCL(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers,
                                   args.ssl_temp, args.ssl_alpha, device),
        'SLMRec': lambda: SLMRec(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.n_layers,
                                 args.ssl_temp, args.ssl_alpha, device),
        'SelfCF': lambda: SelfCF(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers,
                                 args.dropout, device),
        # the one reader of the sampler's second negative (main.py:354-355, dataload.py:81-84)
        'MCLN': lambda: MCLN(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                             args.n_layers, args.n_mca, device),
        # round 5: four more members (main.py:318-320, :342-343, :358-359, :377-378)
        'POWERec': lambda: POWERec(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                                   args.n_layers, int(args.prompt_num), args.neg_weight, args.dropout, device),
        'LGMRec': lambda: LGMRec(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                                 args.n_layers, args.ssl_alpha, device),
        'DHCF': lambda: DHCF(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers,
                             args.dropout, device),
        'FKAN_GCF': lambda: FKAN_GCF(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers,
                                     args.node_dropout, args.message_dropout, args.grid_size, device),
        'LightGT': lambda: LightGT(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                                   args.n_layers, device),
        'MMGCL': lambda: MMGCL(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                               args.n_layers, args.ssl_alpha, args.ssl_temp, args.dropout, device),
        # (main.py:282-283, :314-315)
        'BM3': lambda: BM3(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.feature_embed, args.reg_weight,
                           args.dropout, args.n_layers, args.cl_weight, aggr_mode, device),
        'MGCL': lambda: MGCL(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight, args.n_layers,
                             aggr_mode, args.ssl_temp, args.ssl_alpha, device),
        # (main.py:302-303)
        'SGL': lambda: SGL(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers, aggr_mode,
                           args.ssl_temp, args.ssl_alpha, device),
        # (main.py:309-313)
        'LightGCL': lambda: LightGCL(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers, aggr_mode,
                                     args.ssl_alpha, args.ssl_temp, device),
        'HCCF': lambda: HCCF(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers, aggr_mode,
                             args.ssl_alpha, args.ssl_temp, args.keepRate, args.leaky, args.mult, device),
        # (main.py:346-348)
        'MENTOR': lambda: MENTOR(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.mm_layers, args.reg_weight,
                                 args.ssl_temp, args.dropout, args.align_weight, args.mask_weight_g, args.mask_weight_f, device),
        # (main.py:276-279)
        'LATTICE': lambda: LATTICE(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.feature_embed, args.reg_weight,
                                   args.n_layers, args.mm_layers, args.ii_topk, aggr_mode, args.lambda_coeff, device),
        # (main.py:294-296)
        'MICRO': lambda: MICRO(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.n_layers, args.reg_weight,
                               args.ii_topk, args.mm_layers, args.ssl_temp, args.lambda_coeff, args.ssl_alpha, aggr_mode, device),
        # (main.py:325-326)
        'DCCF': lambda: DCCF(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers, args.ssl_temp,
                             args.ssl_alpha, args.n_intents, args.cen_reg, device),
        # (main.py:299-301)
        'DDRec': lambda: DDRec(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.feature_embed,
                               args.reg_weight, args.n_layers, args.ssl_temp, args.ssl_alpha, args.threshold, aggr_mode, device),
        # (main.py:292-293)
        'MGAT': lambda: MGAT(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight, device),
        # (main.py:271-273)
        'GRCN': lambda: GRCN(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.feature_embed, args.reg_weight,
                             args.dropout, args.n_iterations, aggr_mode, device),
        # (main.py:331-332)
        'MMSSL': lambda: MMSSL(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight, args.ssl_alpha,
                               args.ssl_temp, args.G_rate, args.mm_layers, device),
        # (main.py:333-334)
        'VGCL': lambda: VGCL(num_user, num_item, train_data, user_item_dict, dim_E, args.reg_weight, args.n_layers,
                             args.ssl_temp, args.ssl_alpha, device),
        # (main.py:379-380)
        'GUME': lambda: GUME(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.n_layers,
                             args.n_ui_layers, args.um_loss, args.vt_loss, args.data_path, device),
        'SMORE': lambda: SMORE(num_user, num_item, train_data, user_item_dict, v_feat, t_feat, dim_E, args.reg_weight,
                               args.n_ui_layers, args.ii_topk, args.dropout, args.data_path, device),
    }
    if args.Model not in table:
        raise SystemExit(f"--Model {args.Model}: only {sorted(table)} are on the MI355X hot path")
    return table[args.Model]()


def main(argv=None):
    args = parse_args(argv)
    setup_logging(args)
    logging.info('============Arguments==============')
    for arg, value in vars(args).items():
        logging.info('%s: %s', arg, value)
    logging.info('local time：%s', get_local_time())
    setup_seed(args.seed)
    device = gpu()
    if device.type != "cuda":
        raise SystemExit("chaorec_amd runs on the MI355X only: no GPU visible")
    config = load_yaml_config(args.Model)
    needs_feat = args.Model in ("MMGCN", "FREEDOM", "MGCN", "VBPR", "SLMRec", "MCLN", "POWERec", "LGMRec", "SMORE", "MMGCL", "LightGT", "GUME", "DDRec", "MICRO", "MENTOR", "BM3", "MGCL", "LATTICE", "MMSSL", "GRCN", "MGAT")
    train_data, val_data, test_data, user_item_dict, num_user, num_item, v_feat, t_feat = dataload.data_load(
        args.data_path, has_v=needs_feat, has_t=needs_feat, data_root=args.data_root, synthetic=args.synthetic)
    if args.host_sampler:
        ds = dataload.TrainingDataset(num_user, num_item, user_item_dict, train_data, args.Model)
        train_loader = DataLoader(ds, args.batch_size, shuffle=True, num_workers=args.num_workers)
    else:
        train_loader = dataload.DeviceBatchSampler(num_user, num_item, user_item_dict, train_data, args.batch_size,
                                                   device, args.Model, args.seed)
    args.num_user, args.num_item = num_user, num_item

    combinators = list(product(*[config[p] for p in config['hyper_parameters']]))
    best_performance, best_params, best_metrics = None, None, None
    for idx, combo in enumerate(combinators):
        hyper = dict(zip(config['hyper_parameters'], combo))
        logging.info('========={}/{}: Parameters:{}========='.format(idx + 1, len(combinators), hyper))
        for key, value in hyper.items():
            setattr(args, key, value)
        model = build_model(args, num_user, num_item, train_data, user_item_dict, v_feat, t_feat, device)
        model.to(device)
        # reference: torch.optim.Adam (main.py:397); same update, one fused launch per tensor
        optimizer = FusedAdam([{'params': model.parameters(), 'lr': args.learning_rate}])
        current = train_and_evaluate(model, train_loader, val_data, test_data, optimizer, args.num_epoch,
                                     model_name=args.Model, topk=args.topk, patience=args.patience,
                                     graph=not args.no_graph)
        recall = current[20]['recall'] if 20 in current else current[max(current)]['recall']
        if best_performance is None or recall > best_performance:
            best_performance, best_params, best_metrics = recall, hyper.copy(), current
    logging.info("Best performance: {:.5f}".format(best_performance))
    logging.info("Best parameters: {}".format(best_params))
    logging.info("Best metrics:")
    for k, m in best_metrics.items():
        logging.info(f"{k}: {' | '.join(f'{name}: {value:.5f}' for name, value in m.items())}")
    return best_metrics


if __name__ == '__main__':
    main()
