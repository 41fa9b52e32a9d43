"""ALINK_MTP — the reference's Multi-PIE low-resolution driver (code/ALINK_MTP.py) as a runnable module:

    python -m a_link_amd.ALINK_MTP --dataDirPrefix ../MultiPieSplits/split1/train --testDir ../MultiPieSplits/split1/test

Teacher: VGGFace2 ResNet-50 features at 224 x 224 scored by the ensemble; student: SmallRes trained end
to end on lowRes x lowRes pixels.  Same flags and phases: train the low-res model and quit if it is not
saved yet (code/ALINK_MTP.py:116-125), otherwise run the framework loop (alink_loop.run_alink_mtp),
save, and report top-1 identification on the test split (:271-289).
"""
import argparse
import sys

from . import alink_loop, committee, noise, readDFW, readMTP, siamese

IMAGE_RES, FEATURE_RES = (224, 224), (2048,)


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    alink_loop.add_flags(p)
    p.set_defaults(out_model='MTP_models/postALINK', ensemble_basepath='MTP_models/ensemble', noise='adversarial',
                   batch_send=32, mixture_ratio=1, alink_bs=8, eps=0.1)          # code/ALINK_MTP.py:49-69
    p.add_argument("--dataDirPrefix", default="../MultiPieSplits/split1/train")
    p.add_argument("--testDir", default="../MultiPieSplits/split1/test")
    p.add_argument("--lowres_basemodel", default="MTP_models/lowresModel")
    p.add_argument("--lowRes", type=int, default=48)
    p.add_argument("--lowres_epochs", type=int, default=10)
    p.add_argument("--highres_epochs", type=int, default=5)
    p.add_argument("--resnet50_weights", default=None)
    p.add_argument("--pretrain_steps", type=int, default=32000)
    p.add_argument("--quiet", action="store_true")
    return p


def main(argv=None):
    FLAGS = build_parser().parse_args(argv)
    verbose = 0 if FLAGS.quiet else 1
    low_res = (FLAGS.lowRes, FLAGS.lowRes)
    print("== Low resolution : %s ==" % str(low_res))
    conversionModel = siamese.RESNET50(IMAGE_RES, weights=FLAGS.resnet50_weights)
    X_dig_raw = readMTP.readAllImages(FLAGS.dataDirPrefix)
    assert 0 <= FLAGS.split_ratio <= 1 and 0 <= FLAGS.disparity_ratio <= 1 and 0 <= FLAGS.eps < 0.5
    print("== Noise that will be used for ALINK: %s ==" % (FLAGS.noise))
    if FLAGS.split_ratio > 0:
        (X_dig_pre, X_dig_post) = readDFW.splitDisguiseData(X_dig_raw, pre_ratio=FLAGS.split_ratio)
    else:
        X_dig_pre = X_dig_post = X_dig_raw
    ensemble = [siamese.SiameseNetwork(FEATURE_RES, FLAGS.ensemble_basepath + str(i), 1e-1)
                for i in range(1, FLAGS.num_ensemble_models + 1)]
    lowResModel = siamese.SmallRes(low_res + (3,), FEATURE_RES, FLAGS.lowres_basemodel + str(FLAGS.lowRes), 1e-1)
    ensembleNoise = [noise.get_relevant_noise(x)(model=lowResModel, sess=None, feature_model=None)
                     for x in FLAGS.noise.split(',')]
    bag = committee.Bagging(ensemble, ensembleNoise)
    if not lowResModel.maybeLoadFromMemory():
        print('== Training lowres-faces model ==')
        normGen = readDFW.getNormalGenerator(X_dig_pre, FLAGS.batch_size)
        lowResSiamGen = readMTP.getGenerator(normGen, FLAGS.batch_size, low_res)
        lowResModel.customTrainModel(lowResSiamGen, FLAGS.lowres_epochs, FLAGS.batch_size, 0.2, FLAGS.pretrain_steps,
                                     preprocess=True, verbose=verbose)
        lowResModel.save()
        return None
    print('== Loaded lowres-faces model from memory ==')
    for m in ensemble:
        m.maybeLoadFromMemory()
    normGen = readDFW.getNormalGenerator(X_dig_pre, FLAGS.batch_size)
    dataGen = readMTP.getGenerator(normGen, FLAGS.batch_size, low_res)
    state = alink_loop.run_alink_mtp(FLAGS, conversionModel, bag, ensembleNoise, lowResModel, X_dig_post, dataGen,
                                     IMAGE_RES, low_res, verbose=verbose)
    X_test = readMTP.readAllImages(FLAGS.testDir, low_res)
    state.top1 = alink_loop.top1_identification(lowResModel, X_test)
    print('Top-1 accuracy : ', state.top1)
    return state


if __name__ == "__main__":
    main(sys.argv[1:])
