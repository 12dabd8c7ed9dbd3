"""
Where genomes come from: a directory of ``<id>.json`` files, one JSON document holding many
genomes (``{"type": "onefiledb", id: genome, ...}``), or a flam3 XML file named directly.

One class; it offers the ``get(id)`` that edge blending uses to follow an edge's links to its
nodes (cuburn/genome/blend.py), plus ``animation(name)`` for callers that want something
renderable whatever the stored type is.
"""
import json
import os
import warnings

_XML_EXT = ('.flam3', '.flame')


class GenomeStore(object):
    def __init__(self, path='.'):
        self.path = path
        self.docs = {}                  # ids held in memory: the one-file document, or stashed genomes
        self.directory = None
        if os.path.isdir(path):
            self.directory = path
        else:
            with open(path) as fp:
                doc = json.load(fp)
            if not isinstance(doc, dict) or doc.get('type') != 'onefiledb':
                raise ValueError('%s is neither a directory nor a onefiledb document' % path)
            self.docs.update((k, v) for k, v in doc.items() if k != 'type')

    def stash(self, ident, genome):
        self.docs[ident] = genome

    def get(self, ident):
        if ident in self.docs:
            return self.docs[ident]
        if self.directory is None:
            raise KeyError(ident)
        fn = ident if ident.endswith('.json') else ident + '.json'
        with open(os.path.join(self.directory, fn)) as fp:
            return json.load(fp)

    def animation(self, name, half=False):
        """``(animation genome, basename for output files)`` for an id or a flam3 XML path."""
        from . import convert
        stem, ext = os.path.splitext(os.path.basename(name))
        if ext in _XML_EXT and os.path.isfile(name):
            with open(name) as fp:
                flames = convert.XMLGenomeParser.parse(fp.read())
            if len(flames) != 1:
                warnings.warn('%d flames in file, only using one.' % len(flames))
            genome = convert.flam3_to_node(flames[0])
        else:
            genome = self.get(name)
            if ext not in _XML_EXT + ('.json',):
                stem = os.path.basename(name)
        kind = genome.get('type')
        if kind == 'node':
            genome = convert.node_to_anim(self, genome, half=half)
        elif kind == 'edge':
            genome = convert.edge_to_anim(self, genome)
        elif kind != 'animation':
            raise ValueError('unrecognised genome type %r' % kind)
        return genome, stem


def connect(path):
    return GenomeStore(path)
