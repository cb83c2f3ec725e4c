package embedding;

import java.util.Random;

/**
 * The 48-bit state of a java.util.Random without reflection (JDK 16+ refuses reflective access to java.util).
 * java.util.Random is specified bit for bit (its Javadoc fixes the generator "for the sake of absolute portability"):
 *   state' = (state * 0x5DEECE66D + 0xB) mod 2^48,  next(b) = state' >>> (48 - b),  setSeed(x): state = (x ^ 0x5DEECE66D) mod 2^48.
 * Two nextInt() outputs determine the state (2^16 candidates for the low bits); setSeed puts the generator back where it
 * stood, so reading the state consumes nothing.  The device sampler continues the stream from that state
 * (include/dge.h: dge_sample_walks, rng_mode 0).
 */
final class JavaRandomState {
    static final long MULT = 0x5DEECE66DL, ADD = 0xBL, MASK = (1L << 48) - 1;
    private static final long MULT_INV = 0xDFE05BCB1365L;          // MULT * MULT_INV == 1 mod 2^48

    private JavaRandomState() {}

    /** the state r's next draw starts from; r is left exactly as it was */
    static long peek(Random r) {
        if (r.getClass() != Random.class)
            throw new IllegalStateException("LayeredGraph.rnd must be a plain java.util.Random (is " + r.getClass().getName()
                    + "): the device continues its documented stream");
        synchronized (r) {
            long o1 = r.nextInt() & 0xFFFFFFFFL, o2 = r.nextInt() & 0xFFFFFFFFL;
            for (long lo = 0; lo < 65536; lo++) {
                long s1 = (o1 << 16) | lo;
                long s2 = (s1 * MULT + ADD) & MASK;
                if ((s2 >>> 16) == o2) {
                    long s0 = ((s1 - ADD) * MULT_INV) & MASK;
                    r.setSeed(s0 ^ MULT);
                    return s0;
                }
            }
        }
        throw new IllegalStateException("java.util.Random did not follow its specified generator");
    }

    /** state after `steps` generator steps (one nextDouble = 2 steps), O(log steps) */
    static long jump(long state, long steps) {
        long am = 1, ap = 0, cm = MULT, cp = ADD;
        while (steps != 0) {
            if ((steps & 1) != 0) {
                am = am * cm;
                ap = ap * cm + cp;
            }
            cp = (cm + 1) * cp;
            cm = cm * cm;
            steps >>>= 1;
        }
        return (am * state + ap) & MASK;
    }
}
