"""Pin the model-level CPU oracle (oracle/torch_ref.py) against the reference's own train() iteration
(tests/golden/train_step_{256x512,512x1024}.npz, produced by tests/golden/make_golden_model.py).  CPU only."""
import os

import numpy as np
import pytest
import torch

import model_common as mc


@pytest.mark.parametrize("size", ["256x512", "512x1024"])   # the second is BASELINE.json's configs[1] size
def test_oracle_iteration_matches_reference_train(golden_dir, size):
    g = np.load(os.path.join(golden_dir, "train_step_%s.npz" % size))
    H, W = int(g["H"]), int(g["W"])
    res, (det, dec, dis, dis_patch), _ = mc.oracle_iteration(H, W, lr=float(g["lr"]))
    # logged scalars (the reference prints 5 decimals; accuracies are logged /100)
    got = {k: float(res[k]) for k in ('loss', 'rpn_cls', 'rpn_loc', 'rcnn_cls', 'rcnn_loc', 'fake_loss_target', 'recon_loss',
                                      'adloss', 'fake_loss1_source')}
    got['rpn_acc'] = float(res['rpn_acc'][0]) / 100.
    got['rcnn_acc'] = float(res['rcnn_acc'][0]) / 100.
    for k, v in zip(g["logged_keys"], g["logged"]):
        assert abs(got[str(k)] - float(v)) <= 1.5e-5 + 1e-5 * abs(float(v)), (k, got[str(k)], float(v))
    # post-step state: same seeds + same op order on the same CPU kernels -> agreement to fp32 round-off
    for name, m in (("det", det), ("dec", dec), ("dis", dis), ("dis_patch", dis_patch)):
        np.testing.assert_allclose(mc.si.checksum(m), g["ck_" + name], rtol=1e-7, err_msg=name)
    sd = det.state_dict()
    np.testing.assert_allclose(sd['fc_rcnn_cls.bias'].numpy(), g['det_fc_cls_bias'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(sd['features.0.bias'].numpy(), g['det_conv1_1_bias'], rtol=1e-4, atol=1e-7)
    np.testing.assert_allclose(sd['rpn_head.conv_loc.bias'].numpy(), g['det_rpn_loc_bias'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dis.state_dict()['model_A.3.bias'].numpy(), g['dis_last_bias'], rtol=1e-5, atol=1e-8)
    np.testing.assert_allclose(dec.state_dict()['decode_A.6.bias'].numpy(), g['dec_last_bias_A'], rtol=1e-5, atol=1e-8)
    dp = dis_patch.state_dict()
    np.testing.assert_allclose(dp['model_A_patch.0.model.1.running_mean'].numpy(), g['dp_bn1_running_mean'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dp['model_A_patch.0.model.1.running_var'].numpy(), g['dp_bn1_running_var'], rtol=1e-5, atol=1e-7)
    np.testing.assert_allclose(dp['model_A_patch.0.model.1.weight'].numpy(), g['dp_bn1_weight'], rtol=1e-5, atol=1e-8)
    assert int(dp['model_A_patch.0.model.1.num_batches_tracked']) == 3  # three train-mode forwards per iteration


def test_state_dict_layout_is_the_reference_layout():
    """oracle, product mirror and the reference (SURVEY 2d probe) share key names and shapes"""
    from oracle import torch_ref as R
    from scda_amd.dropin.models.faster_rcnn.vgg_adver_expansion_cluster import vgg16
    from scda_amd.train_step import builder_gan
    o = R.build_models(mc.CFG)
    p_det = vgg16(cfg=dict(mc.CFG['shared'], gan_model_flag=2))
    p_dis, p_dec, p_patch = builder_gan()
    for om, pm in zip(o, (p_det, p_dec, p_dis, p_patch)):
        osd, psd = om.state_dict(), pm.state_dict()
        assert list(osd.keys()) == list(psd.keys())
        for k in osd:
            assert osd[k].shape == psd[k].shape, k
    assert len(p_det.state_dict()) == 40 and sum(p.numel() for p in p_det.parameters()) == 136850887


def test_decoder_32_variant_layout():
    """GAN_decoder_AE_32 (module-API boundary): GAN_decoder_AE behind LinUnsRes_cluster2's bias-free stride-2 conv"""
    from scda_amd.dropin.models.faster_rcnn.faster_rcnn_adver_expansion_reweight_cluster import GAN_decoder_AE, GAN_decoder_AE_32
    p = {'ch': 128, 'input_dim_b': 3, 'n_gen_res_blk': 3, 'n_gen_front_blk': 3, 'res_dropout_ratio': 0.5}
    a, b = GAN_decoder_AE(p).state_dict(), GAN_decoder_AE_32(p).state_dict()
    extra = [k for k in b if k not in a]
    assert extra == ['decode_B.0.model.0.weight', 'decode_A.0.model.0.weight']
    assert tuple(b[extra[0]].shape) == (128, 128, 3, 3) and set(a) <= set(b)
