"""
Arithmetic on (hyper-)parameter distributions: ``Parser(S1, S2, ...)('log(rate2@1*2) + rate@2^2 > 20')``.

Own implementation of the query language of the reference's ``bayesloop.Parser`` (bayesloop/parser.py:50-440; surface:
``Parser(*studies)``, ``parser(query, t=None, silent=False)``, ``Study.eval`` core.py:604-621): a query is an arithmetic
expression over parameter names, hyper-parameter names, numbers and NumPy / scipy.special function names with the operators
``+ - * / ^`` and ``@`` (time stamp selection), optionally followed by ONE relation ``< > <= >= ==`` and a right-hand side.
With a relation the probability of the statement is returned, without it the binned distribution of the derived quantity.

Semantics kept from the reference (file:line refer to bayesloop/parser.py):

* precedence, tightest first: function application, ``@``, ``^`` (right-assoc.), unary sign, ``* /``, ``+ -`` (:121-127);
* a parameter lives on the FULL joint grid of its study (values = ravelled mesh grid, probabilities = ravelled posterior
  of one time step), so two parameters of the same study combine element by element, i.e. with their joint distribution;
  quantities of different studies, the same parameter at two time stamps, hyper-parameter x parameter and derived x derived
  combine as independent variables (outer product of values and probabilities, :229-243); everything else is element-wise
  and keeps the probabilities of the first operand that carries any (NumPy's subclass rule, :246-247);
* a parameter needs a time stamp (``@`` or ``t=``) before arithmetic (:213-223); hyper-parameters of an ``OnlineStudy`` with
  history as well;
* ``lhs REL rhs`` is evaluated as ``-1*(rhs)+lhs REL 0`` (:383-390); the distribution uses
  ``int((max-min)/max gap)`` equal bins labelled by their upper edge (:399-418).

Not kept: the reference prints debugging output while binning (:404-406) and needs ``pyparsing``; only the time steps a query
selects are copied from the GPU (``DevicePosterior.row``) instead of the whole posterior sequence.
"""
from __future__ import annotations

import operator
import re

import numpy as np
import scipy.special as _sp

from .exceptions import ConfigurationError

_ARITH = {'+': operator.add, '-': operator.sub, '*': operator.mul, '/': operator.truediv, '^': operator.pow}
_RELATIONS = (('>=', operator.ge), ('<=', operator.le), ('==', operator.eq), ('>', operator.gt), ('<', operator.lt))
_NUMBER = re.compile(r'(\d+\.?\d*|\.\d+)([eE][+-]?\d+)?')
_NAME = re.compile(r'[A-Za-z_][A-Za-z_0-9]*')


class _Quantity:
    """Values on a grid with their probabilities.  kind: 'param' | 'hyper' | 'derived'; ``rows`` (callable index -> 1-D
    probabilities) stands for the time-resolved probabilities until a time stamp is selected."""

    def __init__(self, values, prob=None, rows=None, name=None, time=None, study=None, kind='param', timestamps=None):
        self.values = np.asarray(values, dtype=float)
        self.prob, self.rows = prob, rows
        self.name, self.time, self.study, self.kind, self.timestamps = name, time, study, kind, timestamps

    def copy(self):
        return _Quantity(self.values.copy(), self.prob, self.rows, self.name, self.time, self.study, self.kind, self.timestamps)

    def with_values(self, values):
        q = self.copy()
        q.values = np.asarray(values, dtype=float)
        return q

    @property
    def needs_time(self):
        return self.rows is not None and self.prob is None

    def select(self, stamp):
        """``quantity @ stamp`` (:203-208)."""
        q = self.copy()
        try:
            index = list(self.timestamps).index(stamp)
        except ValueError:
            raise ValueError('{} is not in list'.format(stamp))
        q.prob, q.rows, q.time = np.asarray(self.rows(index), dtype=float), None, stamp
        return q


def _tokenize(text, names, functions):
    tokens, i = [], 0
    while i < len(text):
        ch = text[i]
        if ch.isspace():
            i += 1
            continue
        if ch in '+-*/^@()':
            tokens.append(('op', ch))
            i += 1
            continue
        m = _NUMBER.match(text, i)
        if m:
            tokens.append(('num', float(m.group(0))))
            i = m.end()
            continue
        m = _NAME.match(text, i)
        if m:
            word = m.group(0)
            if word in names:
                tokens.append(('name', word))
            elif word in functions:
                tokens.append(('func', word))
            else:
                raise ConfigurationError('Unknown name "{}" in query.'.format(word))
            i = m.end()
            continue
        raise ConfigurationError('Cannot parse "{}" in query.'.format(text[i:]))
    return tokens


class Parser:
    """Computes derived probability values and distributions from arithmetic on (hyper-)parameters of one or more studies."""

    def __init__(self, *studies):
        self.studies = studies
        if len(studies) == 0:
            raise ConfigurationError('Parser instance takes at least one Study instance as argument.')
        self.names = []
        for study in studies:
            self.names.extend(study.observationModel.parameterNames)
            if _is_online(study):
                for names in study.hyperParameterNames:
                    self.names.extend(names)
            elif hasattr(study, 'flatHyperParameterNames'):
                self.names.extend(study.flatHyperParameterNames)
        if len(set(self.names)) != len(self.names):
            raise ConfigurationError('Specified study objects contain duplicate parameter names.')
        self.functions = set(n for n in dir(np) if callable(getattr(np, n, None)))
        self.functions |= set(n for n in dir(_sp) if callable(getattr(_sp, n, None)))
        for name in self.names:
            if name in self.functions:
                self.functions.discard(name)
                print('! WARNING: Function "{}" will not be available in parser, as it collides with '
                      '(hyper-)parameter names.'.format(name))

    # ---- quantities of the studies ---------------------------------------------------------------------------------
    def _load(self, t):
        out = {}
        for study in self.studies:
            online = _is_online(study)
            history = getattr(study, 'storeHistory', True)
            stamps = list(study.formattedTimestamps)
            for index, name in enumerate(study.observationModel.parameterNames):
                values = np.ravel(study.grid[index])
                if t is not None:
                    ti = list(self.studies[0].formattedTimestamps).index(t)                 # (:291, first study's stamps)
                    out[name] = _Quantity(values, prob=np.ravel(_posterior_row(study, ti)), name=name, time=t, study=study)
                elif online and not history:
                    out[name] = _Quantity(values, prob=np.ravel(study.marginalizedPosterior), name=name, time=stamps[-1], study=study)
                else:
                    out[name] = _Quantity(values, rows=(lambda i, s=study: np.ravel(_posterior_row(s, i))), name=name, study=study,
                                          timestamps=stamps)
            if online:
                for j, names in enumerate(study.hyperParameterNames):
                    for name in names:
                        k = list(names).index(name)                 # column of this hyper-parameter in the model's value grid
                        values = np.asarray(study.hyperParameterValues[j])[:, k]
                        if t is None and history:
                            seq = study.hyperParameterSequence
                            rows = (lambda i, s=seq, jj=j: np.asarray(s[i][jj]) / np.sum(s[i][jj]))
                            out[name] = _Quantity(values, rows=rows, name=name, study=study, kind='hyper', timestamps=stamps)
                        elif t is None:
                            d = np.asarray(study.hyperParameterDistribution[j], dtype=float)
                            out[name] = _Quantity(values, prob=d / d.sum(), name=name, time=stamps[-1], study=study, kind='hyper')
                        elif history:
                            ti = list(self.studies[0].formattedTimestamps).index(t)
                            d = np.asarray(study.hyperParameterSequence[ti][j], dtype=float)
                            out[name] = _Quantity(values, prob=d / d.sum(), name=name, time=t, study=study, kind='hyper')
                        else:
                            raise ConfigurationError('OnlineStudy instance is not configured to store history, '
                                                     'cannot access t={}.'.format(t))
            elif hasattr(study, 'flatHyperParameterNames'):
                for name in study.flatHyperParameterNames:
                    k = study._getHyperParameterIndex(study.transitionModel, name)
                    d = np.asarray(study.hyperParameterDistribution, dtype=float)
                    grid = getattr(study, 'allHyperGridValues', None)
                    if grid is None or len(grid) == 0:
                        grid = study.hyperGridValues
                    out[name] = _Quantity(np.asarray(grid)[:, k], prob=d / d.sum(), name=name, study=study, kind='hyper')
        return out

    # ---- expression evaluation (precedence climbing) -----------------------------------------------------------------
    def _expression(self, tokens, pos, quantities):
        """sum := product (('+'|'-') product)*"""
        left, pos = self._product(tokens, pos, quantities)
        while pos < len(tokens) and tokens[pos] in (('op', '+'), ('op', '-')):
            sym = tokens[pos][1]
            right, pos = self._product(tokens, pos + 1, quantities)
            left = self._arith(sym, left, right)
        return left, pos

    def _product(self, tokens, pos, quantities):
        left, pos = self._signed(tokens, pos, quantities)
        while pos < len(tokens) and tokens[pos] in (('op', '*'), ('op', '/')):
            sym = tokens[pos][1]
            right, pos = self._signed(tokens, pos + 1, quantities)
            left = self._arith(sym, left, right)
        return left, pos

    def _signed(self, tokens, pos, quantities):
        if pos < len(tokens) and tokens[pos] in (('op', '+'), ('op', '-')):
            sign = -1.0 if tokens[pos][1] == '-' else 1.0
            operand, pos = self._signed(tokens, pos + 1, quantities)
            return self._arith('*', sign, operand), pos                       # "-x" is "(-1)*x" (:152-157)
        return self._power(tokens, pos, quantities)

    def _power(self, tokens, pos, quantities):
        base, pos = self._at(tokens, pos, quantities)
        if pos < len(tokens) and tokens[pos] == ('op', '^'):
            # right-associative; the exponent may carry its own sign ("2^-1")
            exponent, pos = self._signed_power(tokens, pos + 1, quantities)
            return self._arith('^', base, exponent), pos
        return base, pos

    def _signed_power(self, tokens, pos, quantities):
        if pos < len(tokens) and tokens[pos] in (('op', '+'), ('op', '-')):
            sign = -1.0 if tokens[pos][1] == '-' else 1.0
            operand, pos = self._signed_power(tokens, pos + 1, quantities)
            return self._arith('*', sign, operand), pos
        return self._power(tokens, pos, quantities)

    def _at(self, tokens, pos, quantities):
        left, pos = self._applied(tokens, pos, quantities)
        while pos < len(tokens) and tokens[pos] == ('op', '@'):
            stamp, pos = self._applied(tokens, pos + 1, quantities)
            if isinstance(left, _Quantity) and left.needs_time and not isinstance(stamp, _Quantity):
                left = left.select(stamp)
            else:
                raise ConfigurationError('"@" selects a time stamp of a (hyper-)parameter with a history.')
        return left, pos

    def _applied(self, tokens, pos, quantities):
        if pos >= len(tokens):
            raise ConfigurationError('Unexpected end of query.')
        kind, val = tokens[pos]
        if kind == 'func':
            fn = getattr(np, val) if hasattr(np, val) and callable(getattr(np, val)) else getattr(_sp, val)
            operand, pos = self._applied(tokens, pos + 1, quantities)
            if isinstance(operand, _Quantity):
                with np.errstate(all='ignore'):
                    return operand.with_values(fn(operand.values)), pos
            return fn(operand), pos
        if kind == 'num':
            return val, pos + 1
        if kind == 'name':
            return quantities[val].copy(), pos + 1
        if (kind, val) == ('op', '('):
            inner, pos = self._expression(tokens, pos + 1, quantities)
            if pos >= len(tokens) or tokens[pos] != ('op', ')'):
                raise ConfigurationError('Missing ")" in query.')
            return inner, pos + 1
        if (kind, val) in (('op', '+'), ('op', '-')):                        # a sign directly behind a function / "@"
            return self._signed(tokens, pos, quantities)
        raise ConfigurationError('Unexpected "{}" in query.'.format(val))

    def _arith(self, sym, a, b):
        op = _ARITH[sym]
        qa, qb = isinstance(a, _Quantity), isinstance(b, _Quantity)
        for q in (a, b):
            if isinstance(q, _Quantity) and q.needs_time:
                if q.kind == 'hyper':
                    raise ConfigurationError('No timestamp defined for hyper-parameter "{}"'.format(q.name))
                raise ConfigurationError('No timestamp defined for parameter "{}"'.format(q.name))
        with np.errstate(all='ignore'):
            if not qa and not qb:
                return op(a, b)
            if qa and qb and self._independent(a, b):
                values = op(a.values[:, None], b.values[None, :]).ravel()      # (a_i, b_j), a slow (:239-240)
                prob = (np.asarray(a.prob)[:, None] * np.asarray(b.prob)[None, :]).ravel()
                return _Quantity(values, prob=prob / prob.sum(), name='_derived', kind='derived')
            if qa and qb:
                return a.with_values(op(a.values, b.values))                   # joint grid of one study: element by element
            if qa:
                return a.with_values(op(a.values, b))
            return b.with_values(op(a, b.values))

    @staticmethod
    def _independent(a, b):
        pa, pb = a.kind != 'hyper', b.kind != 'hyper'                          # derived quantities are "parameters" (:243)
        if pa and pb:
            return (a.study is not b.study) or (a.study is None and b.study is None) or (a.name == b.name and a.time != b.time)
        if not pa and not pb:
            return (a.study is not b.study) or (a.study is None and b.study is None)
        return True

    # ---- queries -----------------------------------------------------------------------------------------------------
    def __call__(self, query, t=None, silent=False):
        quantities = self._load(t)
        parts = re.split('>=|<=|==|>|<', query)
        if len(parts) > 2:
            raise ConfigurationError('Use exactly one operator out of (<, >, <=, >=, ==) to obtain probability value, '
                                     'or none to obtain derived distribution.')
        reduced = query if len(parts) == 1 else '-1*(' + parts[1] + ')+' + parts[0]
        tokens = _tokenize(reduced, set(self.names), self.functions)
        derived, pos = self._expression(tokens, 0, quantities)
        if pos != len(tokens):
            raise ConfigurationError('Cannot parse query behind "{}".'.format(tokens[pos][1]))
        if not isinstance(derived, _Quantity):
            raise ConfigurationError('Query contains no (hyper-)parameter.')
        if derived.needs_time:
            raise ConfigurationError('No timestamp defined for {}parameter "{}"'.format('hyper-' if derived.kind == 'hyper' else '',
                                                                                        derived.name))
        values, prob = derived.values, np.asarray(derived.prob, dtype=float)

        if len(parts) == 2:
            rel = next(fn for sym, fn in _RELATIONS if sym in query)
            with np.errstate(invalid='ignore'):
                p = float(np.sum(prob[rel(values, 0.)]))
            if not silent:
                print('P({}) = {}'.format(query, p))
            return p

        values = values.copy()
        values[np.isinf(values)] = np.nan
        dmin, dmax = np.nanmin(values), np.nanmax(values)
        gap = np.nanmax(np.diff(np.sort(values)))                               # bin size = largest gap between two derived values
        n_bins = int((dmax - dmin) / gap)
        bins = np.linspace(dmin, dmax, n_bins)
        if not silent:
            print('+ Computing distribution: {}'.format(query))
        # probabilities of the half-open bins [lower, upper), labelled by their upper edge (:408-418)
        index = np.searchsorted(bins, values, side='right') - 1
        ok = ~np.isnan(values) & (index >= 0) & (index < n_bins - 1)
        binned = np.bincount(index[ok], weights=prob[ok], minlength=max(n_bins - 1, 0))[:max(n_bins - 1, 0)]
        return bins[:-1] + (bins[1] - bins[0]), binned


def _is_online(study):
    return hasattr(study, 'transitionModels') and hasattr(study, 'hyperParameterSequence')


def _posterior_row(study, index):
    """One time step of the study's posterior sequence without materialising the sequence when it still lives on the GPU."""
    pending = getattr(study, '_posterior_pending', None)
    if pending is not None and hasattr(pending, 'row'):
        return pending.row(index)
    return np.asarray(study.posteriorSequence[index])
