"""Mirror of src/util.py:ranking — the model factory that is the reference's only plugin API."""


def ranking(FLAGS):
    """util.py:61-96.  Only the models on the accelerated path exist here."""
    if FLAGS.model == "EasyDGL":
        from .model import EasyDGL
        return EasyDGL(FLAGS.num_items, FLAGS)
    if FLAGS.model == "TGAT":
        from .model import TGAT
        return TGAT(FLAGS.num_items, FLAGS)
    if FLAGS.model == "TiSASREC":
        from .model import TiSASRec
        return TiSASRec(FLAGS.num_items, FLAGS)
    if FLAGS.model == "CTSMA":
        from .model import CTSMA
        return CTSMA(FLAGS.num_items, FLAGS)
    raise NotImplementedError("The ranking model: {0} not implemented".format(FLAGS.model))
