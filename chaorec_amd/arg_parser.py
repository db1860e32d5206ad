"""CLI flags and YAML grid loading for the hot-path models (reference arg_parser.py:13-100).

Only the flags LightGCN / MMGCN / FREEDOM and the generic loop read are kept; the other models'
knobs are out of scope.  Unlike the reference, nothing here runs at import time:
dataload.py / train_and_evaluate.py receive `args` explicitly."""
import argparse
import os

import yaml

_HERE = os.path.dirname(os.path.abspath(__file__))


def build_parser():
    parser = argparse.ArgumentParser(description="Run ChaoRec (MI355X hot path).")
    parser.add_argument('--Model', nargs='?', default='LightGCN', help='Model name: LightGCN | MMGCN | FREEDOM | NGCF | MGCN | LayerGCN')
    parser.add_argument('--data_path', nargs='?', default='baby', help='baby, clothing, sports, beauty, microlens, netfilx')
    parser.add_argument('--data_root', default='./Data', help='directory holding <data_path>/train.npy ...')
    parser.add_argument('--learning_rate', type=float, nargs='+', default=1e-3, help='Learning rates')
    parser.add_argument('--feature_embed', type=int, default=64, help='Feature Embedding size')
    parser.add_argument('--batch_size', type=int, default=1024, help='Batch size.')
    parser.add_argument('--aggr_mode', default='add', help='Aggregation mode.')
    parser.add_argument('--reg_weight', type=float, nargs='+', default=1e-3, help='Weight decay.')
    parser.add_argument('--dim_E', type=int, default=64, help='Embedding dimension.')
    parser.add_argument('--num_epoch', type=int, default=1000, help='Epoch number.')
    parser.add_argument('--dropout', type=float, default=0.2, help='Dropout.')
    parser.add_argument('--n_layers', type=int, default=2, help='conv_layers.')
    parser.add_argument('--mm_layers', type=int, default=2, help='the number of multimodal layer.')
    parser.add_argument('--ssl_temp', type=float, default=0.9, help='temperature coefficient.')
    parser.add_argument('--ssl_alpha', type=float, default=0.9, help='ssl coefficient.')
    parser.add_argument('--n_mca', type=int, default=2, help='MCLN counterfactual layer.')
    parser.add_argument('--prompt_num', type=float, default=0.1, help='prompt modal numbers.')
    parser.add_argument('--neg_weight', type=float, default=0.1, help='weak modal weight.')
    parser.add_argument('--n_ui_layers', type=int, default=3, help='n_ui layers.')
    parser.add_argument('--n_iterations', type=int, default=3, help='the number of iteration.')
    parser.add_argument('--G_rate', type=float, default=0.0001, help='MMSSL')
    parser.add_argument('--cl_weight', type=float, default=2.0, help='the number of cl_loss_weight.')
    parser.add_argument('--leaky', type=float, default=0.5, help='HCCF leaky')
    parser.add_argument('--keepRate', type=float, default=1.0, help='HCCF keep rate')
    parser.add_argument('--mult', type=float, default=0.1, help='HCCF hypergraph scale')
    parser.add_argument('--align_weight', type=float, default=0.1, help='MENTOR align_weight')
    parser.add_argument('--mask_weight_f', type=float, default=1.5, help='MENTOR mask_weight_f')
    parser.add_argument('--mask_weight_g', type=float, default=0.001, help='MENTOR mask_weight_g')
    parser.add_argument('--cen_reg', type=float, default=5e-3, help='intent regularization')
    parser.add_argument('--n_intents', type=int, default=128, help='Number of latent intents')
    parser.add_argument('--threshold', type=float, default=0.1, help='the number of threshold.')
    parser.add_argument('--um_loss', type=float, default=0.1, help='um_loss.')
    parser.add_argument('--vt_loss', type=float, default=0.1, help='vt_loss.')
    parser.add_argument('--grid_size', type=int, default=1, help='FKAN_GCF grid_size.')
    parser.add_argument('--node_dropout', type=float, default=0.1, help='FKAN_GCF node_dropout')
    parser.add_argument('--message_dropout', type=float, default=0.1, help='FKAN_GCF message_dropout')
    parser.add_argument('--no_graph', action='store_true',
                        help='eager launches instead of one captured hipGraph replay per training batch')
    parser.add_argument('--ii_topk', type=int, default=10, help='the number of item-item graph topk.')
    parser.add_argument('--lambda_coeff', type=float, default=0.9, help='the number of jump connection factor.')
    parser.add_argument('--seed', type=int, default=42, help='Number of seed')
    parser.add_argument('--num_workers', type=int, default=1, help='Workers number.')
    # reference: type=float (arg_parser.py:92), which breaks ranked_list[:k] for CLI-supplied values; ints here
    parser.add_argument('--topk', type=int, nargs='+', default=[5, 10, 20], help='topK')
    parser.add_argument('--patience', type=int, default=20, help='early-stopping patience (reference: 20)')
    parser.add_argument('--host_sampler', action='store_true',
                        help='use the reference-style DataLoader + python sampler instead of the device sampler')
    parser.add_argument('--synthetic', action='store_true',
                        help='generate a synthetic graph of the dataset shape instead of reading Data/')
    return parser


def parse_args(argv=None):
    return build_parser().parse_args(argv)


def load_yaml_config(model_name):
    yaml_file = os.path.join(_HERE, "Model_YAML", f"{model_name}.yaml")
    with open(yaml_file, 'r') as file:
        return yaml.safe_load(file)
