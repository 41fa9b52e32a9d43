"""ALINK_arc — the reference's ArcFace driver script (code/ALINK_arc.py) as a runnable module:

    python -m a_link_amd.ALINK_arc --dataDirPrefix DFW_Data/ --trainImagesDir Training_data \\
        --arcface_model ./arcface_model/model-r100-ii/model --noise gaussian,saltpepper,poisson,speckle

Same flags, same phases (pre-train / load M2 and the M1 ensemble, then the framework loop
alink_loop.run_alink_dfw), same files written.  `--feature_model resnet50` runs the ALINK.py variant
(VGGFace2 ResNet-50 at 224 x 224, 2048-d features, column 1).  `perlin` cannot run at 112 x 112 in the
reference either and is dropped from the default noise list with a warning.
"""
import argparse
import sys

from . import alink_loop, committee, noise, readDFW, siamese


def build_parser():
    p = argparse.ArgumentParser(description=__doc__.split("\n")[0])
    alink_loop.add_flags(p)
    p.add_argument("--dataDirPrefix", default="DFW_Data/")
    p.add_argument("--trainImagesDir", default="Training_data")
    p.add_argument("--testImagesDir", default="Testing_data")
    p.add_argument("--arcface_model", default="./arcface_model/model-r100-ii/model",
                   help="checkpoint prefix (code/ALINK_arc.py:64) or synthetic:<arch>")
    p.add_argument("--feature_model", default="arcface", choices=["arcface", "resnet50"])
    p.add_argument("--resnet50_weights", default=None, help="keras-vggface weight file for --feature_model resnet50")
    p.add_argument("--pretrain_steps", type=int, default=320000, help="n_steps of customTrainModel (code/siamese.py:81)")
    p.add_argument("--quiet", action="store_true")
    return p


def main(argv=None):
    FLAGS = build_parser().parse_args(argv)
    verbose = 0 if FLAGS.quiet else 1
    if FLAGS.feature_model == "arcface":
        IMAGERES, FEATURERES, col = (112, 112), (512,), 0
        conversionModel = siamese.ArcFace(IMAGERES, FLAGS.arcface_model)
    else:
        IMAGERES, FEATURERES, col = (224, 224), (2048,), 1
        conversionModel = siamese.RESNET50(IMAGERES, weights=FLAGS.resnet50_weights)
    (X_plain, X_dig, X_imp) = readDFW.getAllTrainData(FLAGS.dataDirPrefix, FLAGS.trainImagesDir, IMAGERES, conversionModel)
    (X_plain_raw, X_dig_raw) = readDFW.getRawTrainData(FLAGS.dataDirPrefix, FLAGS.trainImagesDir, IMAGERES)
    assert 0 <= FLAGS.split_ratio <= 1 and 0 <= FLAGS.disparity_ratio <= 1 and 0 <= FLAGS.eps < 0.5
    noises = FLAGS.noise.split(',')
    if IMAGERES[0] % 56 == 0 and IMAGERES[0] // 32 * 32 != IMAGERES[0] and 'perlin' in noises:
        print("perlin noise cannot be generated at %dx%d (the reference's reshape fails too): dropped" % IMAGERES)
        noises = [n for n in noises if n != 'perlin']
    print("Noise that will be used for ALINK: %s" % ",".join(noises))
    if FLAGS.split_ratio > 0:
        (X_dig_pre, _) = readDFW.splitDisguiseData(X_dig, pre_ratio=FLAGS.split_ratio)
        (_, X_dig_post) = readDFW.splitDisguiseData(X_dig_raw, pre_ratio=FLAGS.split_ratio)
    else:
        X_dig_pre, X_dig_post = X_dig, X_dig_raw
    disguisedFacesModel = siamese.SiameseNetwork(FEATURERES, FLAGS.disguised_basemodel, 0.1)
    ensembleNoise = [noise.get_relevant_noise(x)(model=disguisedFacesModel, sess=None, feature_model=conversionModel)
                     for x in noises]
    ensemble = [siamese.SiameseNetwork(FEATURERES, FLAGS.ensemble_basepath + str(i), 0.1)
                for i in range(1, FLAGS.num_ensemble_models + 1)]
    bag = committee.Bagging(ensemble, ensembleNoise)

    def gen_for(X):
        return readDFW.getGenerator(readDFW.getNormalGenerator(X, FLAGS.batch_size),
                                    readDFW.getNormalGenerator(X_imp, FLAGS.batch_size),
                                    readDFW.getImposterGenerator(X, X_imp, FLAGS.batch_size), FLAGS.batch_size, 0)

    if FLAGS.train_disguised_model:
        print('Training disguised-faces model')
        disguisedFacesModel.customTrainModel(gen_for(X_dig_pre), FLAGS.dig_epochs, FLAGS.batch_size, 0.2,
                                             n_steps=FLAGS.pretrain_steps, verbose=verbose)
        disguisedFacesModel.save()
        return None
    disguisedFacesModel.maybeLoadFromMemory()
    print('Loaded disguised-faces model from memory')
    for individualModel in ensemble:
        if alink_loop.pretrain(individualModel, gen_for(X_plain), FLAGS.undig_epochs, FLAGS.batch_size,
                               n_steps=FLAGS.pretrain_steps, refine=FLAGS.refine_models, verbose=verbose):
            print('Finetuned undisguised-faces models')
        else:
            print('Loaded undisguised-faces models')
    return alink_loop.run_alink_dfw(FLAGS, conversionModel, bag, ensembleNoise, disguisedFacesModel, X_plain_raw, X_dig_post,
                                    gen_for(X_plain), IMAGERES, col=col, verbose=verbose)


if __name__ == "__main__":
    main(sys.argv[1:])
