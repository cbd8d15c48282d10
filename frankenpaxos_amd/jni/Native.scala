// Native.scala -- the reference-side binding a frankenpaxos maintainer would add (source only: no
// JDK / scalac in this image).  Drop into jvm/src/main/scala/frankenpaxos/gpu/ of the reference.
//
// `Native` is the JNI surface of frankenpaxos_amd/jni/fpx_jni.c.  `GpuPhase2Engine` owns one libfpx context --
// the acceptors of every group plus the proxy leader's tallies -- and maps the unbounded log onto the
// context's window of rows.  Two thin actors put it behind the unchanged Actor/Transport trait surface:
//   * `GpuProxyLeader` stands at a proxy leader's address: its `receive` only ENQUEUES Phase2a messages, a
//     zero-delay Transport timer (the "tick") flushes the queue through ONE native call and then `send`s the
//     Nack / Chosen messages the Scala handlers would have sent (multipaxos/Acceptor.scala:192-219,
//     multipaxos/ProxyLeader.scala:246-253);
//   * `GpuAcceptor` stands at EVERY acceptor address (all instances share the engine): the Leader's Phase1a
//     (multipaxos/Leader.scala:231, 410-420) is answered with the Phase1b / Nack of
//     multipaxos/Acceptor.scala:148-182, built from fpx_acceptor_phase1a + fpx_acceptor_phase1b_info, which
//     is what the unchanged Leader.handlePhase1b (Leader.scala:504-577) needs to recover and re-propose; and the
//     read path -- MaxSlotRequest / BatchMaxSlotRequest are answered with Acceptor.maxVotedSlot (Acceptor.scala:
//     222-254), which the engine follows per acceptor (GpuPhase2Engine.maxVotedSlot).
package frankenpaxos.gpu

import frankenpaxos.Actor
import frankenpaxos.Chan
import frankenpaxos.Logger
import frankenpaxos.multipaxos._
import frankenpaxos.roundsystem.RoundSystem
import scala.collection.mutable

object Native {
  System.loadLibrary("fpxjni") // libfpxjni.so -> libfpx.so

  // status codes of include/fpx.h
  val OK = 0; val EINVAL = 1; val EFATAL_UNKNOWN_SLOTROUND = 2

  @native def create(cfg: Array[Int]): Long // < 0: -status
  @native def destroy(handle: Long): Int
  @native def acceptorPhase2a(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                              value: Array[Int], targetMask: Array[Long], voteBits: Array[Long],
                              nackBits: Array[Long], nackRound: Array[Int]): Int
  @native def proxyOpen(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                        value: Array[Int], isNew: Array[Byte]): Int
  @native def proxyPhase2b(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                           voteBits: Array[Long], newlyChosen: Array[Byte],
                           chosenRound: Array[Int], chosenValue: Array[Int]): Int
  @native def phase2Fused(handle: Long, n: Int, slot: Array[Int], round: Array[Int],
                          value: Array[Int], targetMask: Array[Long], chosen: Array[Byte],
                          chosenRound: Array[Int], chosenValue: Array[Int], nackRound: Array[Int]): Int
  @native def quorumEval(cfg: Array[Int], n: Int, nodes: Array[Long], strict: Int,
                         out: Array[Byte]): Int

  // page-locked batches: direct buffers over fpx_host_alloc memory (fill through asIntBuffer /
  // asLongBuffer views in ByteOrder.nativeOrder), DMA'd without pinning or copying
  @native def hostAlloc(bytes: Long): java.nio.ByteBuffer // null on failure
  @native def hostFree(buffer: java.nio.ByteBuffer): Int
  @native def phase2FusedDirect(handle: Long, n: Int, slot: java.nio.ByteBuffer,
                                round: java.nio.ByteBuffer, value: java.nio.ByteBuffer,
                                targetMask: java.nio.ByteBuffer, chosen: java.nio.ByteBuffer,
                                chosenRound: java.nio.ByteBuffer, chosenValue: java.nio.ByteBuffer,
                                nackRound: java.nio.ByteBuffer): Int

  // up to 3 ticks in flight on page-locked batches: submit returns a ticket (< 0: -status), wait its status; the
  // PCIe transfers of one tick hide behind the fused step of its neighbours (fpx_phase2_fused_submit / _wait)
  @native def phase2FusedSubmitDirect(handle: Long, n: Int, slot: java.nio.ByteBuffer,
                                      round: java.nio.ByteBuffer, value: java.nio.ByteBuffer,
                                      targetMask: java.nio.ByteBuffer, chosen: java.nio.ByteBuffer,
                                      chosenRound: java.nio.ByteBuffer, chosenValue: java.nio.ByteBuffer,
                                      nackRound: java.nio.ByteBuffer): Int
  @native def phase2FusedWait(handle: Long, ticket: Int): Int

  // the rows around the fused step
  @native def roundLeader(numLeaders: Int, round: Int): Int // < 0: -status (numLeaders < 1)
  @native def acceptorPhase1a(handle: Long, group: Int, round: Int, chosenWatermark: Int,
                              targetMask: Array[Long], bits: Array[Long]): Int
  @native def leaderPhase1bScan(handle: Long, chosenWatermark: Int, numGroups: Int,
                                quorumMasks: Array[Long], cap: Int, maxSlot: Array[Int],
                                safeRound: Array[Int], safeValue: Array[Int]): Int
  @native def replicaChosen(handle: Long, n: Int, slot: Array[Int], value: Array[Int],
                            mask: Array[Byte], state: Array[Int]): Int
  @native def replicaChosenNoopRange(handle: Long, slotStart: Int, slotEnd: Int,
                                     state: Array[Int]): Int
  // Mencius noop ranges: n per call (the fused step), and the unfused pieces for one range
  @native def noopRangesFused(handle: Long, n: Int, numGroups: Int, slotStart: Array[Int],
                              slotEnd: Array[Int], round: Array[Int], targetMasks: Array[Long],
                              voteBits: Array[Long], nackBits: Array[Long], nackRound: Array[Int],
                              isNew: Array[Byte], chosen: Array[Byte]): Int
  @native def acceptorPhase2aNoopRange(handle: Long, slotStart: Int, slotEnd: Int, round: Int,
                                       numGroups: Int, targetMasks: Array[Long],
                                       bits: Array[Long], nackRound: Array[Int]): Int
  @native def proxyOpenNoopRange(handle: Long, slotStart: Int, slotEnd: Int, round: Int,
                                 isNew: Array[Byte]): Int
  @native def proxyPhase2bNoopRange(handle: Long, slotStart: Int, slotEnd: Int, round: Int,
                                    numGroups: Int, voteBits: Array[Long],
                                    newlyChosen: Array[Byte]): Int
  @native def epxCreate(numReplicas: Int, numKeys: Int, device: Int): Long // < 0: -status
  @native def epxDestroy(handle: Long): Int
  @native def epxPreaccept(handle: Long, m: Int, numReplicas: Int, leader: Array[Int],
                           number: Array[Int], key: Array[Int], isSet: Array[Byte],
                           respMask: Array[Byte], seenMask: Array[Byte], rank: Array[Int],
                           fast: Array[Byte], deps: Array[Int], leaderDeps: Array[Int],
                           ownValuesEnd: Array[Int]): Int
  // EPaxos on the command log (Replica.cmdLog): Prepare, Accept, handlePreAccept with ballots / Nacks / re-sent
  // replies.  replies packs the per-message reply bit sets back to back (m bytes each), see fpx_jni.c
  @native def epxCreateWithLog(numReplicas: Int, numKeys: Int, device: Int, numInstances: Int): Long
  @native def epxPrepare(handle: Long, m: Int, numReplicas: Int, leader: Array[Int], number: Array[Int],
                         ballotOrdering: Array[Int], ballotReplica: Array[Int], targetMask: Array[Byte],
                         replies: Array[Byte], nackBallot: Array[Int], prepareOk: Array[Int]): Int
  // Replica.handlePrepareOk: decision = action | source | triple (3 x m); action 0 wait, 1 Accept phase with the
  // triple, 2 pre-accept its command again, 3 pre-accept a Noop; asIntended = 0 evaluates :1810 / :1831 as written
  @native def epxHandlePrepareOks(handle: Long, m: Int, numReplicas: Int, leader: Array[Int], number: Array[Int],
                                  ballotOrdering: Array[Int], ballotReplica: Array[Int], respMask: Array[Byte],
                                  prepareOk: Array[Int], asIntended: Int, decision: Array[Int]): Int
  @native def epxAccept(handle: Long, m: Int, leader: Array[Int], number: Array[Int],
                        ballotOrdering: Array[Int], ballotReplica: Array[Int], tripleId: Array[Int],
                        key: Array[Int], isSet: Array[Byte],
                        targetMask: Array[Byte], replies: Array[Byte], nackBallot: Array[Int]): Int
  // Replica.handleCommit at the replicas of targetMask: deps m x n + depsValuesEnd m, or both null (triple by id alone)
  @native def epxHandleCommit(handle: Long, m: Int, numReplicas: Int, leader: Array[Int], number: Array[Int],
                              tripleId: Array[Int], key: Array[Int], isSet: Array[Byte], deps: Array[Int],
                              depsValuesEnd: Array[Int], targetMask: Array[Byte]): Int
  // Replica.execute on the device: the dependency graph of the committed instances handed in (dense columns first(l) ..
  // first(l) + count(l) - 1), strongly connected components in reverse topological order.  order / component: m each;
  // counts = (executed, components, needsHostPath)
  @native def epxExecute(handle: Long, m: Int, numReplicas: Int, leader: Array[Int], number: Array[Int], deps: Array[Int],
                         depsValuesEnd: Array[Int], committed: Array[Byte], first: Array[Int], count: Array[Int],
                         order: Array[Int], component: Array[Int], counts: Array[Int]): Int
  @native def epxHandlePreaccept(handle: Long, m: Int, numReplicas: Int, leader: Array[Int],
                                 number: Array[Int], ballotOrdering: Array[Int],
                                 ballotReplica: Array[Int], key: Array[Int], isSet: Array[Byte],
                                 tripleId: Array[Int], depsIn: Array[Int], depsInValuesEnd: Array[Int],
                                 targetMask: Array[Byte], replies: Array[Byte], nackBallot: Array[Int],
                                 replyDeps: Array[Int], replyEndTriple: Array[Int]): Int
  @native def epxReadCmdlog(handle: Long, numReplicas: Int, replica: Int, leader: Int, number: Int,
                            entry: Array[Int]): Int
  // multi-GPU: one context per GPU, one RCCL communicator over them (fpx_comm_*); the 128-byte id of
  // commUniqueId travels to the other ranks over the actors' own transport
  @native def commUniqueId(id: Array[Byte]): Int
  @native def commCreate(handle: Long, id: Array[Byte], rank: Int, world: Int): Int
  @native def commDestroy(handle: Long): Int
  // wire adapter: a tick of ProxyLeaderInbound byte arrays packed into one direct buffer + n + 1
  // offsets -> fields = kind | slot | round | isNoop | valueLen | groupIndex | acceptorIndex (7 x n)
  @native def wireDecodeProxyLeaderInbound(buf: java.nio.ByteBuffer, offsets: Array[Long], n: Int,
                                           fields: Array[Int], valueOff: Array[Long],
                                           badIndex: Array[Int]): Int
  // the acceptors' half of Phase 1 and the log window (fpx_acceptor_phase1b_info, fpx_read_acceptor,
  // fpx_recycle_slots, fpx_proxy_forget)
  @native def acceptorPhase1bInfo(handle: Long, group: Int, replica: Int, chosenWatermark: Int, cap: Int,
                                  slot: Array[Int], voteRound: Array[Int], voteValue: Array[Int]): Int // count, < 0: -status
  @native def acceptorRound(handle: Long, group: Int, replica: Int): Int // Acceptor.round, < -1: -status - 1
  // the largest row of [firstRow, firstRow + count) in which the acceptor holds a vote, -1: none, < -1: -status - 1
  @native def acceptorMaxVotedIn(handle: Long, group: Int, replica: Int, firstRow: Int, count: Int): Int
  @native def recycleSlots(handle: Long, firstSlot: Int, count: Int): Int
  @native def proxyForget(handle: Long, firstSlot: Int, count: Int): Int
  // wire adapter, acceptor side: AcceptorInbound bytes -> fields = kind | slot | round | isNoop | valueLen |
  // chosenWatermark (6 x n); LeaderInbound{Phase1b} / {Nack} bytes into a direct buffer (length, < 0: -needed)
  @native def wireDecodeAcceptorInbound(buf: java.nio.ByteBuffer, offsets: Array[Long], n: Int,
                                        fields: Array[Int], valueOff: Array[Long], badIndex: Array[Int]): Int
  @native def wireEncodeLeaderPhase1b(out: java.nio.ByteBuffer, groupIndex: Int, acceptorIndex: Int, round: Int,
                                      nInfo: Int, slot: Array[Int], voteRound: Array[Int],
                                      values: java.nio.ByteBuffer, valueOff: Array[Long], valueLen: Array[Int],
                                      isNoop: Array[Byte]): Long
  @native def wireEncodeLeaderNack(out: java.nio.ByteBuffer, round: Int): Long
  // Mencius on the wire (mencius/Mencius.proto): ProxyLeaderInbound -> fields = kind | slot | slotEnd | round | isNoop |
  // valueLen | groupIndex | acceptorIndex (8 x n); AcceptorInbound -> kind | slot | slotEnd | round | isNoop | valueLen |
  // chosenWatermark (7 x n); LeaderInbound{Nack} is field 7 there.  Phase1b has MultiPaxos' layout: wireEncodeLeaderPhase1b
  @native def wireMenciusDecodeProxyLeaderInbound(buf: java.nio.ByteBuffer, offsets: Array[Long], n: Int,
                                                  fields: Array[Int], valueOff: Array[Long],
                                                  badIndex: Array[Int]): Int
  @native def wireMenciusDecodeAcceptorInbound(buf: java.nio.ByteBuffer, offsets: Array[Long], n: Int,
                                               fields: Array[Int], valueOff: Array[Long], badIndex: Array[Int]): Int
  @native def wireMenciusEncodeLeaderNack(out: java.nio.ByteBuffer, round: Int): Long
  // EPaxos on the wire (epaxos/EPaxos.proto ReplicaInbound): fields = kind | instanceLeader | instanceNumber |
  // ballotOrdering | ballotReplica | replicaIndex | sequenceNumber | voteBallotOrdering | voteBallotReplica | status |
  // isNoop | cmdLen | depsNumReplicas (13 x n); explicit ids of message i: values[valuesOff(i) .. valuesOff(i + 1))
  // (leaders) and the same range behind valuesCap (ids)
  @native def wireEpaxosDecodeReplicaInbound(buf: java.nio.ByteBuffer, offsets: Array[Long], n: Int, maxReplicas: Int,
                                             fields: Array[Int], cmdOff: Array[Long], depsWatermark: Array[Int],
                                             valuesOff: Array[Long], valuesCap: Int, values: Array[Int],
                                             badIndex: Array[Int]): Int
  // head = kind | instanceLeader | instanceNumber | ballotOrdering | ballotReplica | replicaIndex | sequenceNumber |
  // voteBallotOrdering | voteBallotReplica | status | isNoop (11 ints)
  @native def wireEpaxosEncodeReplicaInbound(out: java.nio.ByteBuffer, head: Array[Int], command: java.nio.ByteBuffer,
                                             commandOff: Long, commandLen: Int, numReplicas: Int,
                                             depsWatermark: Array[Int], numValues: Int, values: Array[Int]): Long

  def check(status: Int, logger: Logger): Unit = status match {
    case OK                       => ()
    case EINVAL                   => throw new IllegalArgumentException("libfpx: require failed")
    case EFATAL_UNKNOWN_SLOTROUND =>
      logger.fatal("A ProxyLeader received a Phase2b in a slot and round it never sent a Phase2a in.")
    case s => logger.fatal(s"libfpx status $s")
  }
}

// One libfpx context = the acceptors of every group + the proxy leader's tallies, for a deployment whose
// acceptors and proxy leaders run in ONE process on the GPU box (one Transport event loop: the engine is
// not thread-safe, like every actor, Transport.scala:37-39).
//
// The log window.  The context holds `numSlots` rows; slot s lives in row s % numSlots, the window is
// [base, base + numSlots).  A Phase2a beyond the window waits (`deferred`) until the window has moved; the
// window moves, in chunks, over slots that (a) were chosen through this engine -- Chosen was sent to every
// replica -- and (b) lie at least `retain` slots behind the highest chosen slot: their rows are recycled
// (votes dropped, tallies forgotten, the JVM-side value bytes released).  The reference's acceptors and proxy
// leaders never forget anything (Acceptor.scala:98, ProxyLeader.scala:135); the deviation this buys bounded
// memory with: a replica that lost a Chosen more than `retain` slots ago can no longer recover it from these
// acceptors, so `retain` bounds how far a replica may lag.
class GpuPhase2Engine[Transport <: frankenpaxos.Transport[Transport]](
    logger: Logger,
    config: Config[Transport],
    numSlots: Int = 1 << 20,
    retainSlots: Int = 1 << 18,
    // thrifty delivery (the reference's default, ProxyLeader.scala:190-191: every Phase2a goes to f + 1 of its group's
    // 2f + 1 acceptors).  The reference shuffles; any f + 1 will do, and this engine sends to a window of f + 1
    // NEIGHBOURING acceptors that rotates from message to message -- see thriftyMask
    thrifty: Boolean = true
) {
  config.checkValid()
  val perGroup: Int = config.acceptorAddresses(0).size
  val numGroups: Int = config.numAcceptorGroups
  // rows keep their acceptor group when the log wraps (slot % numGroups == row % numGroups)
  logger.check(config.flexible || numSlots % numGroups == 0)
  private val chunk = math.max(1, numSlots / 16)
  logger.check(retainSlots + 2 * chunk <= numSlots)

  private val cfg: Array[Int] =
    if (!config.flexible)
      Array(numSlots, perGroup, numGroups, 1, config.f, /*THRESHOLD*/ 0, 0, 0,
            config.numLeaders, /*ACCEPTOR*/ 0, 4, 0, 0, 0, 0)
    else
      Array(numSlots, numGroups * perGroup, 1, 1, config.f, /*GRID*/ 2,
            numGroups, perGroup, config.numLeaders, 0, 4, 0, 0, 0, 0)
  private val handle = Native.create(cfg)
  if (handle < 0) Native.check((-handle).toInt, logger)

  // where acceptor (groupIndex, index) of the reference's addressing lives in the context
  def ctxGroup(groupIndex: Int): Int = if (config.flexible) 0 else groupIndex
  def ctxReplica(groupIndex: Int, index: Int): Int = if (config.flexible) groupIndex * perGroup + index else index

  // ---- value ids: the int32 the GPU carries stands for a CommandBatchOrNoop kept here; Noop is FPX_NOOP = -1.
  // An id lives as long as the row it was proposed in: votes and tallies of the row are its only holders.
  private val values = mutable.ArrayBuffer[CommandBatchOrNoop]()
  private val freeIds = mutable.ArrayStack[Int]()
  private val idsOfRow = Array.fill(numSlots)(List.empty[Int])
  private def intern(row: Int, v: CommandBatchOrNoop): Int =
    if (v.value.isNoop) -1
    else {
      val id = if (freeIds.nonEmpty) freeIds.pop() else { values += null; values.size - 1 }
      values(id) = v
      idsOfRow(row) = id :: idsOfRow(row)
      id
    }
  def valueOf(id: Int): CommandBatchOrNoop =
    if (id < 0) CommandBatchOrNoop().withNoop(Noop()) else values(id)

  // ---- the window
  private var base = 0                                   // first slot of the window (a multiple of chunk)
  private val chosenInWindow = new java.util.BitSet(numSlots) // by row
  private var chosenPrefix = 0                           // every slot in [base, chosenPrefix) is chosen
  private var highestChosen = -1
  private val deferred = mutable.Queue[Phase2a]()
  private def row(slot: Int): Int = slot % numSlots
  private def slotOfRow(r: Int): Int = base + ((r - row(base)) % numSlots + numSlots) % numSlots

  private def markChosen(slot: Int): Unit = {
    chosenInWindow.set(row(slot))
    highestChosen = math.max(highestChosen, slot)
    while (chosenPrefix < base + numSlots && chosenInWindow.get(row(chosenPrefix))) chosenPrefix += 1
  }
  // moves the window over chosen slots that are old enough; true if it moved
  private def advanceWindow(): Boolean = {
    var moved = false
    while (chosenPrefix - base >= chunk && highestChosen - (base + chunk) >= retainSlots) {
      val r0 = row(base)                                 // chunk | numSlots: the chunk does not wrap
      refreshStaleGroups()                               // (the votes of these rows are read before they are cleared)
      Native.check(Native.recycleSlots(handle, r0, chunk), logger)
      for (r <- r0 until r0 + chunk) {
        idsOfRow(r).foreach(id => { values(id) = null; freeIds.push(id) })
        idsOfRow(r) = Nil
        chosenInWindow.clear(r)
      }
      base += chunk
      moved = true
    }
    moved
  }

  // ---- the read path: Acceptor.maxVotedSlot (multipaxos/Acceptor.scala:104, 208) of every acceptor, in SLOTS.  The
  // device keeps the scalar over rows, which stops being the maximum over slots once the window has wrapped; so the
  // engine follows it here.  A message nobody Nacked was voted for by every acceptor it went to (Acceptor.scala:201-219):
  // the tick's outputs say that much without any per-acceptor answer.  After a tick with a Nack the engine does not know
  // which of the message's acceptors voted; the group is marked stale and the next read asks the device for the largest
  // voted row of each lap of the window (fpx_acceptor_max_voted_in) -- exact again, and only after leader changes.
  private val ctxGroups = if (config.flexible) 1 else numGroups
  private val ctxReplicas = if (config.flexible) numGroups * perGroup else perGroup
  private val maxVoted = Array.fill(ctxGroups, ctxReplicas)(-1)
  private val maxVotedStale = Array.fill(ctxGroups)(false)
  private def groupOfSlot(slot: Int): Int = if (config.flexible) 0 else slot % numGroups

  // after a tick: `slot` = the messages' slots (log positions, not rows), masks = their targets (null: everyone)
  private def noteVotes(slots: Array[Int], nackRound: Array[Int], masks: Array[Long]): Unit = {
    // descending by slot: an acceptor's maximum is the first Nack-free message that reached it
    val order = slots.indices.sortBy(i => -slots(i).toLong)
    val covered = Array.fill(ctxGroups)(new java.util.BitSet(ctxReplicas))
    for (i <- order) {
      val g = groupOfSlot(slots(i))
      if (nackRound(i) >= 0) maxVotedStale(g) = true
      else if (covered(g).cardinality < ctxReplicas) {
        for (a <- 0 until ctxReplicas
             if !covered(g).get(a) && (masks == null || (masks(4 * i + (a >> 6)) & (1L << (a & 63))) != 0)) {
          covered(g).set(a)
          maxVoted(g)(a) = math.max(maxVoted(g)(a), slots(i))
        }
      }
    }
  }

  // the rescan of one stale group: the largest voted row of each lap of the window, per acceptor, from the device
  private def refreshMaxVoted(g: Int): Unit = {
    val r0 = row(base)
    for (b <- 0 until ctxReplicas) {
      // the older lap of the window lives in rows [r0, numSlots), the newer one in [0, r0)
      val hi = if (r0 > 0) Native.acceptorMaxVotedIn(handle, g, b, 0, r0) else -1
      val lo = if (hi < 0) Native.acceptorMaxVotedIn(handle, g, b, r0, numSlots - r0) else -1
      if (hi < -1) Native.check(-hi - 1, logger)
      if (lo < -1) Native.check(-lo - 1, logger)
      val slot = if (hi >= 0) base + (numSlots - r0) + hi else if (lo >= 0) base + (lo - r0) else -1
      maxVoted(g)(b) = math.max(maxVoted(g)(b), slot)
    }
    maxVotedStale(g) = false
  }
  // ADVICE r05: a stale group is rescanned BEFORE any of its rows is recycled (advanceWindow) -- a vote cast in a Nacked
  // message must be recorded before its row is cleared, or MaxSlotReply.slot would under-report (Acceptor.scala:208 never
  // misses a vote, and a low slot is the unsafe direction for linearizable reads)
  private def refreshStaleGroups(): Unit =
    for (g <- 0 until ctxGroups if maxVotedStale(g)) refreshMaxVoted(g)

  // Acceptor.handleMaxSlotRequest / handleBatchMaxSlotRequest reply with this (Acceptor.scala:222-254)
  def maxVotedSlot(groupIndex: Int, index: Int): Int = {
    val g = ctxGroup(groupIndex); val a = ctxReplica(groupIndex, index)
    if (maxVotedStale(g)) refreshMaxVoted(g)
    maxVoted(g)(a)
  }

  // ---- which acceptors a Phase2a goes to.  A window of f + 1 neighbouring acceptors: on groups of 253 .. 256 acceptors it
  // moves in steps of 16 (a 64-byte sector of the group's row in HBM) and does not wrap over the end of the row -- the
  // runs libfpx walks two rows per wavefront step (include/fpx.h, FPX_F_SCATTERED_TARGETS); on the small groups of an
  // everyday deployment it moves by one acceptor and wraps, so every acceptor sees (f + 1) / (2f + 1) of the messages
  private var rotor = 0
  private def thriftyMask(masks: Array[Long], at: Int): Unit = {
    val q = config.f + 1
    val start = if (perGroup >= 253) 16 * (rotor % ((perGroup - q) / 16 + 1)) else rotor % perGroup
    rotor += 1
    for (j <- 0 until q) {
      val a = (start + j) % perGroup
      masks(4 * at + (a >> 6)) |= 1L << (a & 63)
    }
  }

  // ---- Phase 2: one tick of Phase2a messages (ProxyLeader.handlePhase2a + every Acceptor.handlePhase2a +
  // ProxyLeader.handlePhase2b).  Returns, in message order, Chosen to broadcast and (round, Nack) to route.
  case class TickResult(chosen: Seq[Chosen], nacks: Seq[(Int, Nack)])
  def phase2Tick(incoming: Seq[Phase2a]): TickResult = {
    val chosenOut = mutable.Buffer[Chosen](); val nackOut = mutable.Buffer[(Int, Nack)]()
    var batch: Seq[Phase2a] = deferred.dequeueAll(_ => true) ++ incoming
    while (batch.nonEmpty) {
      // below the window: chosen long ago and recycled -- nothing left to vote on; beyond it: wait
      val (now, later) = batch.filter(_.slot >= base).partition(_.slot < base + numSlots)
      val n = now.size
      val slot = new Array[Int](n); val round = new Array[Int](n); val value = new Array[Int](n)
      for ((p, i) <- now.zipWithIndex) {
        slot(i) = row(p.slot); round(i) = p.round; value(i) = intern(row(p.slot), p.commandBatchOrNoop)
      }
      val chosen = new Array[Byte](n); val cr = new Array[Int](n); val cv = new Array[Int](n)
      val nr = new Array[Int](n)
      // thrifty: one mask of f + 1 acceptors per message (ProxyLeader.scala:190-191); a flexible deployment's grid
      // column (:193-196) is not generated here: dense delivery (targetMask = null), as without `thrifty`
      val masks: Array[Long] =
        if (thrifty && !config.flexible && perGroup > config.f + 1) {
          val m = new Array[Long](4 * n)
          for (i <- 0 until n) thriftyMask(m, i)
          m
        } else null
      if (n > 0) Native.check(Native.phase2Fused(handle, n, slot, round, value, masks, chosen, cr, cv, nr), logger)
      noteVotes(now.map(_.slot).toArray, nr, masks)
      for (i <- 0 until n) {
        if (chosen(i) != 0) {
          chosenOut += Chosen(slot = now(i).slot, commandBatchOrNoop = valueOf(cv(i))) // ProxyLeader.scala:246-253
          markChosen(now(i).slot)
        }
        // Acceptor.scala:197-198: Nack(round = acceptor's round) to leaders(roundSystem.leader(phase2a.round))
        if (nr(i) >= 0) nackOut += ((round(i), Nack(round = nr(i))))
      }
      // a moved window may admit what waited; otherwise it keeps waiting for the slots before it to be chosen
      batch = if (advanceWindow()) later else { deferred ++= later; Seq.empty }
    }
    TickResult(chosenOut, nackOut)
  }

  // ---- Phase 1, acceptor side (Acceptor.handlePhase1a, multipaxos/Acceptor.scala:148-182)
  def handlePhase1a(groupIndex: Int, index: Int, phase1a: Phase1a): Either[Nack, Phase1b] = {
    val g = ctxGroup(groupIndex); val a = ctxReplica(groupIndex, index)
    val target = new Array[Long](4); target(a >> 6) = 1L << (a & 63)
    val bits = new Array[Long](8)
    // rows, not slots: a watermark inside the window is a row boundary only when the window does not wrap
    // between it and the window's end -- promise from row 0 on (more than asked for is safe, Acceptor.scala
    // promises on its single `round` anyway) and filter the info by slot below
    Native.check(Native.acceptorPhase1a(handle, g, phase1a.round, 0, target, bits), logger)
    if ((bits(4 + (a >> 6)) & (1L << (a & 63))) != 0) {
      // :155-162  phase1a.round < round: Nack(round)
      return Left(Nack(round = Native.acceptorRound(handle, g, a)))
    }
    // :163-181  Phase1b(info = votes in slots >= chosenWatermark, ascending)
    var cap = 1024
    var slots = new Array[Int](cap); var vr = new Array[Int](cap); var vv = new Array[Int](cap)
    var k = Native.acceptorPhase1bInfo(handle, g, a, 0, cap, slots, vr, vv)
    if (k > cap) {
      cap = k; slots = new Array[Int](cap); vr = new Array[Int](cap); vv = new Array[Int](cap)
      k = Native.acceptorPhase1bInfo(handle, g, a, 0, cap, slots, vr, vv)
    }
    if (k < 0) Native.check(-k, logger)
    val info = (0 until k)
      .map(j => Phase1bSlotInfo(slot = slotOfRow(slots(j)), voteRound = vr(j), voteValue = valueOf(vv(j))))
      .filter(_.slot >= phase1a.chosenWatermark)
      .sortBy(_.slot)
    Right(Phase1b(groupIndex = groupIndex, acceptorIndex = index, round = phase1a.round, info = info))
  }

  // a Phase2a sent straight to one acceptor (not how the reference's Leader sends them, but part of
  // AcceptorInbound): Acceptor.handlePhase2a, multipaxos/Acceptor.scala:184-220
  def handlePhase2a(groupIndex: Int, index: Int, p: Phase2a): Either[Nack, Phase2b] = {
    logger.check(p.slot >= base && p.slot < base + numSlots)
    val a = ctxReplica(groupIndex, index)
    val target = new Array[Long](4); target(a >> 6) = 1L << (a & 63)
    val votes = new Array[Long](4); val nacks = new Array[Long](4); val nr = new Array[Int](1)
    Native.check(Native.acceptorPhase2a(handle, 1, Array(row(p.slot)), Array(p.round),
                                        Array(intern(row(p.slot), p.commandBatchOrNoop)), target, votes, nacks, nr),
                 logger)
    if (nr(0) >= 0) Left(Nack(round = nr(0)))
    else {
      val g = ctxGroup(groupIndex)
      maxVoted(g)(a) = math.max(maxVoted(g)(a), p.slot)                               // Acceptor.scala:208
      Right(Phase2b(groupIndex = groupIndex, acceptorIndex = index, slot = p.slot, round = p.round))
    }
  }

  def close(): Unit = Native.check(Native.destroy(handle), logger)
}

// Stands where a ProxyLeader stands (ProxyLeaderMain): Leaders keep sending Phase2a to it exactly as they send
// to a ProxyLeader (multipaxos/Leader.scala:364-398); replicas keep receiving Chosen from it.
class GpuProxyLeader[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    engine: GpuPhase2Engine[Transport]
) extends Actor(address, transport, logger) {
  override type InboundMessage = ProxyLeaderInbound
  override val serializer = ProxyLeaderInboundSerializer

  private val roundSystem = new RoundSystem.ClassicRoundRobin(config.numLeaders)
  private val leaders = for (a <- config.leaderAddresses) yield chan[Leader[Transport]](a, Leader.serializer)
  private val replicas = for (a <- config.replicaAddresses) yield chan[Replica[Transport]](a, Replica.serializer)
  private val pending = mutable.Buffer[Phase2a]()

  // one tick: a zero-delay timer, i.e. "after the messages already queued on the event loop"
  private val tick = timer("gpuPhase2Tick", java.time.Duration.ZERO, () => flushTick())

  override def receive(src: Transport#Address, inbound: ProxyLeaderInbound): Unit = {
    inbound.request match {
      case ProxyLeaderInbound.Request.Phase2A(p) =>
        if (pending.isEmpty) tick.start()
        pending += p
      case ProxyLeaderInbound.Request.Phase2B(_) =>
        logger.fatal("GpuProxyLeader tallies on the device; it never receives Phase2b messages.")
      case ProxyLeaderInbound.Request.Empty =>
        logger.fatal("Empty ProxyLeaderInbound encountered.")
    }
  }

  private def flushTick(): Unit = {
    val result = engine.phase2Tick(pending.toList)
    pending.clear()
    for (c <- result.chosen) replicas.foreach(_.send(ReplicaInbound().withChosen(c)))
    for ((round, nack) <- result.nacks)
      leaders(roundSystem.leader(round)).send(LeaderInbound().withNack(nack))
  }
}

// Stands at ONE acceptor address (AcceptorMain); every acceptor address of the deployment gets one, all over the
// same engine.  The Leader's Phase1a goes to acceptor addresses (multipaxos/Leader.scala:410-420): without an
// actor here a new leader would never finish Phase 1.
class GpuAcceptor[Transport <: frankenpaxos.Transport[Transport]](
    address: Transport#Address,
    transport: Transport,
    logger: Logger,
    config: Config[Transport],
    engine: GpuPhase2Engine[Transport]
) extends Actor(address, transport, logger) {
  override type InboundMessage = AcceptorInbound
  override val serializer = AcceptorInboundSerializer

  logger.check(config.acceptorAddresses.flatten.contains(address))
  private val groupIndex = config.acceptorAddresses.indexWhere(_.contains(address))
  private val index = config.acceptorAddresses(groupIndex).indexOf(address)
  private val roundSystem = new RoundSystem.ClassicRoundRobin(config.numLeaders)

  override def receive(src: Transport#Address, inbound: AcceptorInbound): Unit = {
    inbound.request match {
      case AcceptorInbound.Request.Phase1A(phase1a) =>
        val leader = chan[Leader[Transport]](src, Leader.serializer)
        engine.handlePhase1a(groupIndex, index, phase1a) match {
          case Left(nack)     => leader.send(LeaderInbound().withNack(nack))        // Acceptor.scala:155-162
          case Right(phase1b) => leader.send(LeaderInbound().withPhase1B(phase1b))  // Acceptor.scala:163-181
        }
      case AcceptorInbound.Request.Phase2A(phase2a) =>
        engine.handlePhase2a(groupIndex, index, phase2a) match {
          case Left(nack) =>                                                          // Acceptor.scala:192-199
            val leader = chan[Leader[Transport]](config.leaderAddresses(roundSystem.leader(phase2a.round)),
                                                 Leader.serializer)
            leader.send(LeaderInbound().withNack(nack))
          case Right(phase2b) =>                                                      // Acceptor.scala:211-219
            chan[ProxyLeader[Transport]](src, ProxyLeader.serializer)
              .send(ProxyLeaderInbound().withPhase2B(phase2b))
        }
      case AcceptorInbound.Request.MaxSlotRequest(r) =>                               // Acceptor.scala:222-237
        chan[Client[Transport]](src, Client.serializer).send(
          ClientInbound().withMaxSlotReply(
            MaxSlotReply(commandId = r.commandId, groupIndex = groupIndex, acceptorIndex = index,
                         slot = engine.maxVotedSlot(groupIndex, index))))
      case AcceptorInbound.Request.BatchMaxSlotRequest(r) =>                          // Acceptor.scala:239-254
        chan[ReadBatcher[Transport]](src, ReadBatcher.serializer).send(
          ReadBatcherInbound().withBatchMaxSlotReply(
            BatchMaxSlotReply(readBatcherIndex = r.readBatcherIndex, readBatcherId = r.readBatcherId,
                              acceptorIndex = index, slot = engine.maxVotedSlot(groupIndex, index))))
      case AcceptorInbound.Request.Empty =>
        logger.fatal("Empty AcceptorInbound encountered.")
    }
  }
}
